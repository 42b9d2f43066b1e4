// AD + 9x7 census cost volume with fused WTA -- replaces class AD_Census
// (AD-CensusV1/AD-Census.h) for both views.
//
// Formulation (different from the reference's per-(i,j,d) census rebuild, AD-Census.h:142-269,
// same results):
//   left  view: census(i,j,d) = popcount((cenA_L[i][j] ^ cenX_R[i][max(j-d,-3)]) & M[i][j])
//               AD(i,j,d)     = |L[i][j] - R[i][max(j-d,0)]|
//   right view: census(i,j,d) = popcount((cenA_R[i][j] ^ cenX_L[i][min(j+d,W+3)]) & M[i][j])
//               AD(i,j,d)     = |L[i][min(j+d,W-1)] - R[i][j]|
// with cenA_* the ordinary 63-bit census of the anchor image (invalid taps = 0), M the
// in-image tap mask of the anchor pixel (validity is tested on anchor coordinates,
// AD-Census.h:173 / :238), cenX_R the right image's census on a left-replicate-extended
// row (:159-160, :177-178) and cenX_L the left image's census with the reference's
// "column 0 past the right edge" rule for neighbours and "column W-1" rule for the
// centre (:224-225, :242-243).  The "copy d-1" branches of ComputeAD (:88-92, :116-120)
// are exactly the clamps above.
//   cost = lutA[AD] + lutC[census], lutA[k] = 1-expf(-(k/sigmaC)), lutC[k] = 1-expf(-(k/sigmaS))
// built on the HOST with the reference's own float expression (:287-289), so the device
// never evaluates exp.  AD in 0..255 and census in 0..63 because images are integer-valued.
//
// Main kernel: one wavefront spans the disparity axis of one pixel; lane l owns the C =
// D/64 consecutive hypotheses d = l*C .. l*C+C-1 and stores them with one
// dword/x2/x3/x4 store (64*C*4 contiguous bytes per wave).  Row operands are staged in
// LDS once per workgroup; WTA is a wave min + ballot (first strict minimum, :355-373).
#include "smt_common.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <new>
#include <type_traits>

namespace {

constexpr int TJ = 64;        // pixels per workgroup
constexpr int NT = 256;       // threads per workgroup (4 waves)

struct Tables {
    // per view v (0 = left anchor, 1 = right anchor)
    uint64_t *cenA[2];   // [H][W]
    uint64_t *cenX[2];   // [H][WX]   view0: index x+3, x in [-3,W-1]; view1: index x, x in [0,W+3]
    uint64_t *mask;      // [H][W]
    uint8_t *u8[2];      // [H][W] left, right as bytes
    float *lut;          // 256 + 64 floats
    int *flag;           // domain flag
    int WX;
    unsigned long long *stamp;   // diagnostics only (smt_adcensus_diag): 4 counters per workgroup, else null
};

// ---- tap validity mask + the four census tables --------------------------------------
// bit layout: tap t = (r+4)*7 + (c+3), r in [-4,4] outer, c in [-3,3] inner, MSB first:
// bit (62 - t)  (63 left shifts of a 64-bit word, AD-Census.h:171-172).
//
// Away from the left/right borders the extended tables equal the ordinary censuses:
//   cenX_R[xr] == cenA_R[xr] for xr >= 3      (no neighbour column is clamped to 0)
//   cenX_L[xl] == cenA_L[xl] for xl <= W-4    (no neighbour wraps to column 0, centre unclamped)
// so k_prep computes the two ordinary censuses + mask once per pixel from an LDS-staged 64x32 tile and
// writes them to both places; prep_edges (extra workgroups of the same launch) fills the 6 + 7 special
// columns per row.
constexpr int PTW = 64;                                     // tile: 64 columns x 32 rows per workgroup
constexpr int PTH = 32;
constexpr int PNT = 256;                                    // each thread does one column of 8 rows

__device__ __forceinline__ unsigned to_u8_checked(float a, bool &bad)
{
    const int ia = (int)a;
    bad = bad || !(a >= 0.0f && a <= 255.0f && (float)ia == a);
    return (unsigned)ia & 0xffu;
}

// the border columns of the extended tables: xr in [-3, 2] and xl in [W-3, W+3], by the general rule.
// Runs in extra workgroups of the k_prep launch (16 rows each, 13 of 16 lanes per row), reading the
// float images directly (their domain is checked by the tile workgroups).
__device__ __forceinline__ void prep_edges(const float *__restrict__ Lf, const float *__restrict__ Rf, int H, int W,
                                           const Tables &T, int block)
{
    const int i = block * (PNT / 16) + (threadIdx.x >> 4);
    const int e = threadIdx.x & 15;
    if (i >= H || e >= 13) return;
    const int WX = T.WX;
    if (e < 6) {
        const int xr = e - 3;
        if (xr > W - 1) return;                          // narrower than 3 columns
        const int cc = xr < 0 ? 0 : xr;
        const unsigned rc = (unsigned)(int)Rf[(size_t)i * W + cc];
        uint64_t w = 0;
#pragma unroll 1
        for (int r = -4; r <= 4; r++) {
            const int ii = i + r;
            const bool rv = (ii >= 0 && ii < H);
            const int ic = ii < 0 ? 0 : (ii > H - 1 ? H - 1 : ii);
#pragma unroll
            for (int c = -3; c <= 3; c++) {
                int jj = xr + c;
                jj = jj < 0 ? 0 : jj;                    // left replicate (:177-178)
                const bool v = rv && jj < W;
                const unsigned val = (unsigned)(int)Rf[(size_t)ic * W + (jj < W ? jj : W - 1)];
                w = (w << 1) | (uint64_t)(v && rc > val);
            }
        }
        T.cenX[0][(size_t)i * WX + e] = w;
    } else {
        const int xl = W - 3 + (e - 6);
        if (xl < 0) return;
        const int cc = xl > W - 1 ? W - 1 : xl;          // centre clamps to W-1 (:224-225)
        const unsigned lc = (unsigned)(int)Lf[(size_t)i * W + cc];
        uint64_t w = 0;
#pragma unroll 1
        for (int r = -4; r <= 4; r++) {
            const int ii = i + r;
            const bool rv = (ii >= 0 && ii < H);
            const int ic = ii < 0 ? 0 : (ii > H - 1 ? H - 1 : ii);
#pragma unroll
            for (int c = -3; c <= 3; c++) {
                int jj = xl + c;
                jj = jj >= W ? 0 : jj;                   // neighbour wraps to column 0 (:242-243)
                const bool v = rv && jj >= 0;
                const unsigned val = (unsigned)(int)Lf[(size_t)ic * W + (jj < 0 ? 0 : jj)];
                w = (w << 1) | (uint64_t)(v && lc > val);
            }
        }
        T.cenX[1][(size_t)i * WX + xl] = w;
    }
}

// Census words of PRR consecutive rows of one column of one image (IMG 0 = left, 1 = right) from the staged tile.
// The two images are processed one after the other and a thread's rows in groups of PRR, so the live window is
// (PRR + 8) x 2 dwords and only PRR rows' store addresses are in flight: with PRR = 1 prep_tile takes 51 VGPRs (122 in
// the form that kept 16 rows of both images in registers) and fits beside the cost kernel it is fused with (below) at
// 8 waves per SIMD.
constexpr int PSC = PTW + 6;                                // staged columns (3-column halo)
constexpr int PSW = (PSC + 2 + 3) / 4 + 1;                  // row stride in dwords (one spare for the 3rd dword)
constexpr int PRR = 1;
template <int IMG>
__device__ __forceinline__ void prep_rows(const uint32_t (*__restrict__ sw)[PSW], int i0, int row0, int x, int col,
                                          unsigned colbits, int H, int W, const Tables &T)
{
    // The thread's 7-byte window [col, col+6] of each of the PRR+8 staged rows it needs, as two dwords: three
    // aligned LDS dwords shifted by col%4 bytes.  Bytes past col+6 are never used.
    const int cw = col >> 2;
    const unsigned sh = (unsigned)(col & 3);
    uint32_t w0[PRR + 8], w1[PRR + 8];
#pragma unroll
    for (int r = 0; r < PRR + 8; r++) {
        const uint32_t a0 = sw[row0 + r][cw], a1 = sw[row0 + r][cw + 1], a2 = sw[row0 + r][cw + 2];
        w0[r] = __builtin_amdgcn_alignbyte(a1, a0, sh); w1[r] = __builtin_amdgcn_alignbyte(a2, a1, sh);
    }
#pragma unroll
    for (int rr = 0; rr < PRR; rr++) {
        const int i = i0 + row0 + rr;
        if (i >= H) break;
        // centre = byte 3 of the window of staged row rr+4
        const int cc = (int)(w0[rr + 4] >> 24);
        T.u8[IMG][(size_t)i * W + x] = (uint8_t)cc;
        // census word, MSB first over taps t = r*7 + c: bit = centre > neighbour = sign of
        // (neighbour - centre), shifted in with one v_alignbit per tap.  Taps 0..30 fill the high word
        // (bits 62..32), taps 31..62 the low word; every staged byte is readable, the raw bits are
        // masked afterwards with the tap-validity word.
        uint32_t ch = 0, cl = 0;
        uint64_t m = 0;
#pragma unroll
        for (int r = 0; r < 9; r++) {
            const int ii = i + r - 4;
            if (ii >= 0 && ii < H) m |= (uint64_t)colbits << (56 - 7 * r);
#pragma unroll
            for (int c = 0; c < 7; c++) {
                const int t = r * 7 + c;
                const uint32_t w = c < 4 ? w0[rr + r] : w1[rr + r];
                const int dv = (int)((w >> (8 * (c & 3))) & 0xffu) - cc;
                if (t < 31) ch = __builtin_amdgcn_alignbit(ch, (uint32_t)dv, 31);
                else cl = __builtin_amdgcn_alignbit(cl, (uint32_t)dv, 31);
            }
        }
        const uint64_t cen = (((uint64_t)ch << 32) | cl) & m;
        const size_t p = (size_t)i * W + x;
        T.cenA[IMG][p] = cen;
        if (IMG == 0) {
            T.mask[p] = m;
            if (x <= W - 4) T.cenX[1][(size_t)i * T.WX + x] = cen;      // cenX_L index = xl
        } else {
            if (x >= 3) T.cenX[0][(size_t)i * T.WX + x + 3] = cen;      // cenX_R index = xr + 3
        }
    }
}

// One workgroup of the table launch: (bx, by) of a (gdx, ...) grid -- a 64 x 32 tile, or past the tile rows a block
// of border columns.  Called with workgroup-uniform arguments by k_prep and by the table workgroups of k_cost_fast2p.
__device__ __forceinline__ void prep_tile(const float *__restrict__ Lf, const float *__restrict__ Rf, int H, int W,
                                          const Tables &T, int bx, int by, int gdx)
{
    constexpr int SR = PTH + 8;                             // staged rows (4-row halo)
    __shared__ uint32_t sLw[SR][PSW];
    __shared__ uint32_t sRw[SR][PSW];
    const int tiles_y = (H + PTH - 1) / PTH;
    if (by >= tiles_y) {                                    // workgroup-uniform, before any barrier
        prep_edges(Lf, Rf, H, W, T, (by - tiles_y) * gdx + bx);
        return;
    }
    const int i0 = by * PTH;
    const int x0 = bx * PTW;
    const int tid = threadIdx.x;
    // stage the tile + halo straight from the float images (coordinates clamped; out-of-image taps
    // are masked); this also is the float -> u8 conversion + domain check
    bool bad = false;
#pragma unroll 2
    for (int e = tid; e < SR * PSC; e += PNT) {
        const int r = e / PSC, c = e - r * PSC;
        int ii = i0 + r - 4, jj = x0 + c - 3;
        ii = ii < 0 ? 0 : (ii > H - 1 ? H - 1 : ii);
        jj = jj < 0 ? 0 : (jj > W - 1 ? W - 1 : jj);
        ((uint8_t *)sLw[r])[c] = (uint8_t)to_u8_checked(Lf[(size_t)ii * W + jj], bad);
        ((uint8_t *)sRw[r])[c] = (uint8_t)to_u8_checked(Rf[(size_t)ii * W + jj], bad);
    }
    if (__syncthreads_or(bad) && tid == 0) atomicOr(T.flag, 1);
    const int col = tid & (PTW - 1);
    const int x = x0 + col;
    if (x >= W) return;
    constexpr int RPT = PTH / (PNT / PTW);                  // output rows per thread (consecutive)
    static_assert(RPT % PRR == 0, "");
    // column validity of the 7 taps, MSB = leftmost tap
    unsigned colbits = 0;
#pragma unroll
    for (int c = 0; c < 7; c++) colbits |= (unsigned)(x + c - 3 >= 0 && x + c - 3 < W) << (6 - c);
#pragma unroll 1
    for (int s = 0; s < RPT / PRR; s++) {
        const int row0 = (tid / PTW) * RPT + s * PRR;
        prep_rows<0>(sLw, i0, row0, x, col, colbits, H, W, T);
        __builtin_amdgcn_sched_barrier(0);                  // one image's window at a time in registers
        prep_rows<1>(sRw, i0, row0, x, col, colbits, H, W, T);
    }
}

__global__ void __launch_bounds__(PNT) k_prep(const float *__restrict__ Lf, const float *__restrict__ Rf,
                                              int H, int W, Tables T)
{
    prep_tile(Lf, Rf, H, W, T, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

template <int C> struct vecf { float v[C]; };          // C = 5..8 (D > 256): plain struct, the kernels there use the per-element paths
template <> struct vecf<1> { float v[1]; };
template <> struct __attribute__((aligned(8))) vecf<2> { float v[2]; };
template <> struct vecf<3> { float v[3]; };
template <> struct __attribute__((aligned(16))) vecf<4> { float v[4]; };

// ---- cost volume + WTA ---------------------------------------------------------------
// grid: (ceil(W/TJ), H, nviews)   block: 256
// FULL: D == 64*C (every lane active, vector store); otherwise C = ceil(D/64) with tail
// lanes masked and scalar stores.
template <int C, bool FULL>
__global__ void __launch_bounds__(NT) k_cost(int H, int W, int D, Tables T, int view0,
                                             float *__restrict__ vol0, float *__restrict__ vol1,
                                             float *__restrict__ disp0, float *__restrict__ disp1)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int view = view0 + blockIdx.z;
    const int i = blockIdx.y;
    const int j0 = blockIdx.x * TJ;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int NX = TJ + D;                                  // staged ext entries (>= TJ + D - 1)

    uint64_t *s_cenx = (uint64_t *)smem;                    // NX
    uint64_t *s_cena = s_cenx + NX;                         // TJ
    uint64_t *s_mask = s_cena + TJ;                         // TJ
    float *s_lutA = (float *)(s_mask + TJ);                 // 256
    float *s_lutC = s_lutA + 256;                           // 64
    uint8_t *s_valx = (uint8_t *)(s_lutC + 64);             // NX
    uint8_t *s_vala = s_valx + NX;                          // TJ

    const int WX = T.WX;
    const uint64_t *cenX = T.cenX[view] + (size_t)i * WX;
    const uint8_t *extv = T.u8[view ^ 1] + (size_t)i * W;   // left view reads R, right view reads L
    const uint8_t *ancv = T.u8[view] + (size_t)i * W;
    // staged entry e <-> ext coordinate x:  view 0: x = j0 - (D-1) + e ; view 1: x = j0 + e
    const int xbase = (view == 0) ? (j0 - (D - 1)) : j0;
    for (int e = tid; e < NX; e += NT) {
        int x = xbase + e;
        int xc, xv;
        if (view == 0) {
            xc = x < -3 ? -3 : (x > W - 1 ? W - 1 : x);
            xv = xc < 0 ? 0 : xc;
            xc += 3;
        } else {
            xc = x > W + 3 ? W + 3 : x;
            xv = x > W - 1 ? W - 1 : x;
        }
        s_cenx[e] = cenX[xc];
        s_valx[e] = extv[xv];
    }
    for (int e = tid; e < TJ; e += NT) {
        int j = j0 + e;
        if (j > W - 1) j = W - 1;
        s_cena[e] = T.cenA[view][(size_t)i * W + j];
        s_mask[e] = T.mask[(size_t)i * W + j];
        s_vala[e] = ancv[j];
    }
    for (int e = tid; e < 320; e += NT) s_lutA[e] = T.lut[e];
    __syncthreads();

    float *vol = view == 0 ? vol0 : vol1;
    float *disp = view == 0 ? disp0 : disp1;
    const int dl = lane * C;                                // first hypothesis of this lane

    for (int p = wid; p < TJ; p += NT / 64) {
        const int j = j0 + p;
        if (j >= W) break;
        const uint64_t ca = s_cena[p], mk = s_mask[p];
        const int va = s_vala[p];
        // ext entry of hypothesis d: view 0: e = p + (D-1) - d ; view 1: e = p + d
        float c[C];
        float best = 0.0f; int bestk = 0;
#pragma unroll
        for (int k = 0; k < C; k++) {
            const int d = dl + k;
            int e = (view == 0) ? (p + (D - 1) - d) : (p + d);
            if (!FULL) e = (d < D) ? e : 0;
            const uint64_t x = (ca ^ s_cenx[e]) & mk;
            const int hd = __popcll(x);
            int ad = va - (int)s_valx[e];
            ad = ad < 0 ? -ad : ad;
            const float cost = s_lutA[ad] + s_lutC[hd];
            c[k] = cost;
            if (k == 0) { best = cost; bestk = 0; }
            else if (best > cost) { best = cost; bestk = k; }
        }
        float *out = vol + ((size_t)i * W + j) * D + dl;
        if (FULL) {
            vecf<C> pk;
#pragma unroll
            for (int k = 0; k < C; k++) pk.v[k] = c[k];
            *reinterpret_cast<vecf<C> *>(out) = pk;
        } else {
#pragma unroll
            for (int k = 0; k < C; k++)
                if (dl + k < D) out[k] = c[k];
        }
        if (disp) {
            float bv = best;
            if (!FULL) {
                // lanes past D must never win; lanes straddling D keep only their valid prefix
                if (dl >= D) bv = INFINITY;
                else if (dl + C > D) {
                    bv = c[0]; bestk = 0;
#pragma unroll
                    for (int k = 1; k < C; k++)
                        if (dl + k < D && bv > c[k]) { bv = c[k]; bestk = k; }
                }
            }
            const int wd = wave_argmin_first(bv, dl + bestk);
            if (lane == 0) disp[(size_t)i * W + j] = (float)wd;
        }
    }
}

// ---- production path (any D <= 256; FULL specialisation for D == 64*C) -------------------
// Same mapping (one wave per pixel, lane l owns d = l*C..l*C+C-1) with the instruction count
// cut down: VIEW is a template parameter; each wave walks FPW CONSECUTIVE pixels so that
// a lane's C ext operands form a register sliding window (one new LDS entry per pixel
// instead of C); values are staged pre-multiplied by 4 so v_sad_u16 yields the lutA byte
// address directly; the WTA is a DPP min over the cost bit patterns (costs are >= +0, so
// uint order == float order) + ballot + scalar pick, no LDS round trips; the FPW winners
// are collected one per lane and stored once per wave.
#ifndef SMT_FTJ
#define SMT_FTJ 64
#endif
constexpr int FTJ = SMT_FTJ;      // pixels per workgroup
constexpr int FPW = FTJ / 4;      // consecutive pixels per wave

// Streaming (non-temporal) store of C consecutive floats: the volumes are written once and are far
// larger than L2 + Infinity Cache, so they should not displace the tables the next workgroups read.
// With the XCD-contiguous chunk order this is worth 5-6 % of the kernel (A/B); with chunks in
// dispatch order it cost 4 %.
// The volume stores are buffer stores: SGPR resource (base = the wave's first pixel, range = the wave's run, so a
// store past the run is dropped by the hardware) + a constant VGPR lane offset + an SGPR offset that advances by
// one pixel (D * 4 bytes) with a scalar add -- no vector instruction is spent on store addresses.  aux = 2 is the
// nt bit of gfx940+ (what __builtin_nontemporal_store puts on a global_store).
typedef int si2 __attribute__((ext_vector_type(2)));
typedef int si3 __attribute__((ext_vector_type(3)));
typedef int si4 __attribute__((ext_vector_type(4)));
template <int C, bool NTS = true>
__device__ __forceinline__ void st_stream(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const float (&v)[C])
{
    constexpr int AUX = NTS ? 2 : 0;
    if (C == 1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v[0]), r, voff, soff, AUX);
    else if (C == 2) {
        si2 x; x.x = __float_as_int(v[0]); x.y = __float_as_int(v[C > 1 ? 1 : 0]);
        __builtin_amdgcn_raw_buffer_store_b64(x, r, voff, soff, AUX);
    } else if (C == 3) {
        si3 x; x.x = __float_as_int(v[0]); x.y = __float_as_int(v[C > 1 ? 1 : 0]); x.z = __float_as_int(v[C > 2 ? 2 : 0]);
        __builtin_amdgcn_raw_buffer_store_b96(x, r, voff, soff, AUX);
    } else {
        si4 x; x.x = __float_as_int(v[0]); x.y = __float_as_int(v[C > 1 ? 1 : 0]); x.z = __float_as_int(v[C > 2 ? 2 : 0]);
        x.w = __float_as_int(v[C > 3 ? 3 : 0]);
        __builtin_amdgcn_raw_buffer_store_b128(x, r, voff, soff, AUX);
    }
}

struct __attribute__((aligned(16))) Anchor { uint64_t cen, mask; };

// FULL: D == 64*C (no lane / element predication anywhere).  Otherwise C = ceil(D/64): lanes whose
// first hypothesis is >= D compute a harmless duplicate of lane 0, elements past D are neither
// stored nor allowed to win the WTA, and the staged arrays carry XPAD spare entries on both sides for
// the window slots those elements would touch.
template <int C, int VIEW, bool FULL, bool NTS = true>
__device__ __forceinline__ void cost_fast_body(int H, int W, int Drt, const Tables &T, float *__restrict__ vol,
                                               float *__restrict__ disp, int i, int bx)
{
    constexpr int DM = 64 * C;                               // largest D this instantiation serves
    constexpr int XPAD = 4;
    constexpr int NXM = FTJ + DM + 2 * XPAD;
    const int D = FULL ? DM : Drt;
    const int NX = FTJ + D;
    __shared__ Anchor s_anc[FTJ];
    __shared__ uint64_t s_cenx_raw[NXM];
    __shared__ float s_lut[320];
    __shared__ uint16_t s_valx_raw[NXM];
    __shared__ uint16_t s_vala[FTJ];
    uint64_t *s_cenx = s_cenx_raw + XPAD;                    // entry e in [-XPAD, NX + XPAD)
    uint16_t *s_valx = s_valx_raw + XPAD;

    const int j0 = bx * FTJ;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    {
        const uint64_t *cenX = T.cenX[VIEW] + (size_t)i * T.WX;
        const uint8_t *extv = T.u8[VIEW ^ 1] + (size_t)i * W;
        const uint8_t *ancv = T.u8[VIEW] + (size_t)i * W;
        const int xbase = (VIEW == 0) ? (j0 - (D - 1)) : j0;
        for (int e = tid - (FULL ? 0 : XPAD); e < NX + (FULL ? 0 : XPAD); e += NT) {
            const int x = xbase + e;
            int xc, xv;
            if (VIEW == 0) {
                xc = x < -3 ? -3 : (x > W - 1 ? W - 1 : x);
                xv = xc < 0 ? 0 : xc;
                xc += 3;
            } else {
                xc = x > W + 3 ? W + 3 : (x < 0 ? 0 : x);
                xv = x > W - 1 ? W - 1 : (x < 0 ? 0 : x);
            }
            s_cenx[e] = cenX[xc];
            s_valx[e] = (uint16_t)(4u * extv[xv]);
        }
        for (int e = tid; e < FTJ; e += NT) {
            int j = j0 + e;
            if (j > W - 1) j = W - 1;
            Anchor a;
            a.cen = T.cenA[VIEW][(size_t)i * W + j];
            a.mask = T.mask[(size_t)i * W + j];
            s_anc[e] = a;
            s_vala[e] = (uint16_t)(4u * ancv[j]);
        }
        for (int e = tid; e < 320; e += NT) s_lut[e] = T.lut[e];
    }
    __syncthreads();

    const int p0 = wid * FPW;
    const int dlr = lane * C;                                // first hypothesis of this lane
    const int dl = (FULL || dlr < D) ? dlr : 0;              // lanes entirely past D shadow lane 0
    bool ok[C];
#pragma unroll
    for (int k = 0; k < C; k++) ok[k] = FULL || (dlr + k < D);
    // ext entry of (pixel p, hypothesis dl+k):  VIEW 0: p + (D-1) - dl - k ;  VIEW 1: p + dl + k
    // With E(n) = staged entry e0 + n, pixel q needs  VIEW 0: E(q-k)  /  VIEW 1: E(q+k), k = 0..C-1.
    // The C live entries sit in a register ring R[n mod C] = E(n); the pixel loop is unrolled by C so
    // every ring index is a compile-time constant (no register shuffling), and one new entry is
    // fetched per pixel.
    const int e0 = (VIEW == 0) ? (p0 + (D - 1) - dl) : (p0 + dl);
    uint64_t rc[C];
    unsigned rv[C];
#pragma unroll
    for (int k = 0; k < C; k++) {
        // ring slot of E(n) is n mod C;  VIEW 0 starts with E(0), E(-1), ..., E(-(C-1));  VIEW 1 with E(0..C-1)
        const int n = (VIEW == 0) ? -k : k;
        const int slot = ((n % C) + C) % C;
        rc[slot] = s_cenx[e0 + n];
        rv[slot] = s_valx[e0 + n];
    }
    const char *lutA = (const char *)s_lut;
    const float *lutC = s_lut + 256;
    const unsigned ooff = (unsigned)dlr * 4u;
    unsigned osoff = 0;                                      // scalar byte offset of the current pixel in the wave's run
    int res = 0;
    const int npx = min(FPW, W - (j0 + p0));                 // uniform; may be <= 0
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(vol + ((size_t)i * W + j0 + p0) * D), 0, (npx > 0 ? npx : 0) * D * 4, 0x00020000);
    // interior run: every pixel of this wave has all 63 taps inside the image, so the tap mask is
    // all ones and the two ANDs per hypothesis can be dropped (bit 63 is 0 in every table entry)
    const bool interior = (i >= 4) && (i < H - 4) && (j0 + p0 >= 3) && (j0 + p0 + npx - 1 <= W - 4);

    auto run = [&](auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        // The anchor entries are read at wave-uniform LDS addresses.  The two addresses live in VGPRs that advance
        // once per group of C pixels (the pixels of a group use immediate offsets); left to itself the compiler keeps
        // them in SGPRs and spends two v_mov per pixel to feed the ds_read.
        typedef const __attribute__((address_space(3))) uint64_t *lds_u64_p;
        typedef const __attribute__((address_space(3))) uint16_t *lds_u16_p;
        uint32_t anc_a = (uint32_t)(size_t)(const __attribute__((address_space(3))) Anchor *)(s_anc + p0);
        uint32_t val_a = (uint32_t)(size_t)(const __attribute__((address_space(3))) uint16_t *)(s_vala + p0);
        asm volatile("" : "+v"(anc_a), "+v"(val_a));
        for (int g = 0; g < npx; g += C) {
#pragma unroll
            for (int u = 0; u < C; u++) {
                const int q = g + u;
                if (q < npx) {
                    Anchor a;
                    if (MASKED) {                                            // one ds_read_b128
                        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                        const u64x2 v = *(const __attribute__((address_space(3))) u64x2 *)(size_t)(anc_a + 16u * u);
                        a.cen = v.x; a.mask = v.y;
                    }
                    else { a.cen = *(lds_u64_p)(size_t)(anc_a + 16u * u); a.mask = 0; }
                    const unsigned va = *(lds_u16_p)(size_t)(val_a + 2u * u);
                    // entry that joins the ring for the next pixel (always inside the staged range)
                    const int nn = (VIEW == 0) ? (q + 1) : (q + C);
                    const uint64_t nc = s_cenx[e0 + nn];
                    const unsigned nv = s_valx[e0 + nn];
                    float c[C];
                    unsigned key[C];
#pragma unroll
                    for (int k = 0; k < C; k++) {
                        // E(q-k) / E(q+k) with q = u (mod C)
                        constexpr int dummy = 0; (void)dummy;
                        const int slot = (VIEW == 0) ? (((u - k) % C) + C) % C : (u + k) % C;
                        uint64_t x = a.cen ^ rc[slot];
                        if (MASKED) x &= a.mask;
                        const int hd = __popcll(x);
                        const unsigned ad4 = __builtin_amdgcn_sad_u16(va, rv[slot], 0u);   // 4*|va - vx|
                        c[k] = *(const float *)(lutA + ad4) + lutC[hd];
                        key[k] = ok[k] ? __float_as_uint(c[k]) : 0xFFFFFFFFu;
                    }
                    if (FULL) st_stream<C, NTS>(orsrc, ooff, osoff, c);
                    else {
#pragma unroll
                        for (int k = 0; k < C; k++)
                            if (ok[k]) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(c[k]), orsrc, ooff + 4u * k, osoff, NTS ? 2 : 0);
                    }
                    osoff += (unsigned)D * 4u;
                    if (disp) {
                        unsigned ml = key[0];
#pragma unroll
                        for (int k = 1; k < C; k++) ml = min(ml, key[k]);
                        const unsigned m = wave_min_u32(ml);
                        const unsigned long long b = __ballot(ml == m);
                        const int first = __builtin_ctzll(b);
                        int kk = C - 1;
#pragma unroll
                        for (int k = C - 2; k >= 0; k--)
                            if ((unsigned)__builtin_amdgcn_readlane((int)key[k], first) == m) kk = k;
                        const int wd = first * C + kk;       // wave-uniform
                        res = (lane == q) ? wd : res;
                    }
                    // the new entry replaces the one that just left the window
                    const int ns = (VIEW == 0) ? (u + 1) % C : u % C;
                    rc[ns] = nc; rv[ns] = nv;
                }
            }
            anc_a += 16u * C; val_a += 2u * C;
        }
    };
    if (interior) run(std::false_type{});
    else run(std::true_type{});
    if (disp && lane < npx) disp[(size_t)i * W + j0 + p0 + lane] = (float)res;
}

// Workgroup -> 64-pixel chunk.  The chunks of a launch form one linear space (view, row, chunk-in-row)
// and workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md), so XCD x takes the x-th contiguous eighth of
// that space: every XCD then streams its own region of the volumes instead of every eighth 48 KB piece
// of the same region.  A store-only kernel with this mapping writes 10 % faster (6.9 vs 6.25 TB/s,
// same box) than with chunks in dispatch order.  The grid is 1-D, padded to a multiple of 8.
__device__ __forceinline__ bool chunk_of_block(int nbx, int H, int nviews, int &view, int &i, int &bx, long b = -1)
{
    const long nb = (long)nbx * H * nviews;
    const long per = (nb + 7) >> 3;
    if (b < 0) b = blockIdx.x;
    const long c = (b & 7) * per + (b >> 3);
    if (c >= nb) return false;
    const long rows = (long)nbx * H;
    view = (int)(c / rows);
    const long r = c - (long)view * rows;
    i = (int)(r / nbx);
    bx = (int)(r - (long)i * nbx);
    return true;
}

template <int C, int VIEW, bool FULL>
__global__ void __launch_bounds__(NT) k_cost_fast(int H, int W, int D, Tables T, float *__restrict__ vol,
                                                  float *__restrict__ disp, int nbx)
{
    int view, i, bx;
    if (!chunk_of_block(nbx, H, 1, view, i, bx)) return;
    cost_fast_body<C, VIEW, FULL>(H, W, D, T, vol, disp, i, bx);
}

// both views in one launch (no gap / tail between two launches)
template <int C, bool FULL, bool NTS = true>
__global__ void __launch_bounds__(NT) k_cost_fast2(int H, int W, int D, Tables T, float *__restrict__ vol0,
                                                   float *__restrict__ vol1, float *__restrict__ disp0,
                                                   float *__restrict__ disp1, int nbx)
{
    int view, i, bx;
    if (!chunk_of_block(nbx, H, 2, view, i, bx)) return;
    // diagnostic launches only (T.stamp != null, smt_adcensus_diag): the shader-clock and the 100 MHz
    // real-time counters around the first wave's work; their ratio is the clock the kernel ran at
    // (MI355X_MICROARCH.md, DVFS item 6).  Ordinary launches take the null branch and execute no stamp.
    unsigned long long t0 = 0, r0 = 0;
    if (T.stamp) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if (view == 0) cost_fast_body<C, 0, FULL, NTS>(H, W, D, T, vol0, disp0, i, bx);
    else cost_fast_body<C, 1, FULL, NTS>(H, W, D, T, vol1, disp1, i, bx);
    if (T.stamp && threadIdx.x == 0) {
        unsigned long long *s = T.stamp + 4 * (size_t)blockIdx.x;
        s[0] = t0; s[1] = r0; s[2] = __builtin_amdgcn_s_memtime(); s[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// k_cost_fast2 of pair n with the table workgroups of pair n + 1 spread through its grid (smt_adcensus_compute_batch):
// one launch per pair on one stream -- no second stream, no events between the table and cost kernels -- with the
// table workgroups (VALU / LDS work) running beside the store-bound cost workgroups instead of in front of them.
// Workgroups come in groups of 8 (one per XCD, so a cost workgroup keeps the XCD of its chunk, chunk_of_block): after
// every `every` cost groups one group of table workgroups, tile (p % ptx, p / ptx) of the table launch for the next
// pair's images, writing the OTHER table set (Tn); what does not fit that pattern follows at the end of the grid.
// fused_grid() is the host's side of the same arithmetic.  Measured against the two-stream form (tables of pair
// n + 1 on an internal stream, two event edges per pair): DESIGN.md section 4.
struct FusedGrid { int cgroups, pgroups, every, slots, groups; };
__host__ __device__ inline FusedGrid fused_grid(int ncost, int nprep)
{
    FusedGrid f;
    f.cgroups = ncost >> 3;                                  // ncost is a multiple of 8
    f.pgroups = (nprep + 7) >> 3;
    f.every = f.cgroups / (f.pgroups > 0 ? f.pgroups : 1);
    if (f.every < 1) f.every = 1;
    f.slots = f.cgroups / f.every;                           // table groups inside the cost range (>= pgroups unless every == 1)
    f.groups = f.cgroups + (f.pgroups > f.slots ? f.pgroups : f.slots);
    return f;
}
// group g of the fused grid -> (true, table group) or (false, cost group)
__host__ __device__ inline bool fused_decode(const FusedGrid &f, int g, int &idx)
{
    const int period = f.every + 1, inter = f.slots * period;
    if (g < inter) {
        const int q = g / period, r = g - q * period;
        if (r == f.every) { idx = q; return true; }
        idx = q * f.every + r;
        return false;
    }
    const int g2 = g - inter, crest = f.cgroups - f.slots * f.every;
    if (g2 < crest) { idx = f.slots * f.every + g2; return false; }
    idx = f.slots + (g2 - crest);
    return true;
}

template <int C, bool FULL, bool NTS = true>
__global__ void __launch_bounds__(NT, 8) k_cost_fast2p(int H, int W, int D, Tables T, float *__restrict__ vol0,
                                                       float *__restrict__ vol1, float *__restrict__ disp0,
                                                       float *__restrict__ disp1, int nbx, int ncost, int nprep,
                                                       const float *__restrict__ nL, const float *__restrict__ nR, Tables Tn,
                                                       int ptx)
{
    static_assert(PNT == NT, "");
    const FusedGrid f = fused_grid(ncost, nprep);
    int idx;
    const bool table = fused_decode(f, (int)(blockIdx.x >> 3), idx);   // workgroup-uniform
    const int b = idx * 8 + (int)(blockIdx.x & 7);
    if (table) {
        if (b < nprep) prep_tile(nL, nR, H, W, Tn, b % ptx, b / ptx, ptx);
        return;
    }
    int view, i, bx;
    if (!chunk_of_block(nbx, H, 2, view, i, bx, b)) return;
    if (view == 0) cost_fast_body<C, 0, FULL, NTS>(H, W, D, T, vol0, disp0, i, bx);
    else cost_fast_body<C, 1, FULL, NTS>(H, W, D, T, vol1, disp1, i, bx);
}

// Store-only twin of k_cost_fast2<C, true>: the same grid, workgroup -> chunk order and streaming stores
// of 64*C*4 bytes per pixel-wave, no tables, no arithmetic.  What it reaches on the handle's own volumes
// is the ceiling the memory system gives this store pattern in this process (smt_adcensus_diag).
template <int C>
__global__ void __launch_bounds__(NT) k_store_only2(int H, int W, float *__restrict__ vol0,
                                                    float *__restrict__ vol1, int nbx)
{
    constexpr int D = 64 * C;
    int view, i, bx;
    if (!chunk_of_block(nbx, H, 2, view, i, bx)) return;
    float *vol = view ? vol1 : vol0;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j0 = bx * FTJ, p0 = wid * FPW;
    const int npx = min(FPW, W - (j0 + p0));
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(vol + ((size_t)i * W + j0 + p0) * D), 0, (npx > 0 ? npx : 0) * D * 4, 0x00020000);
    const unsigned ooff = (unsigned)(lane * C) * 4u;
    unsigned osoff = 0;
    float x[C];
#pragma unroll
    for (int k = 0; k < C; k++) x[k] = (float)(lane + k);
    for (int q = 0; q < npx; q++) {
        st_stream<C>(orsrc, ooff, osoff, x);
        osoff += (unsigned)D * 4u;
        x[0] += 1.0f;
    }
}

// WTA over an existing volume: one wave per pixel, lane owns C consecutive d (one vector load when
// D == 64*C), DPP argmin.
template <int C, bool FULL>
__global__ void __launch_bounds__(NT) k_wta(const float *__restrict__ vol, int N, int D,
                                            float *__restrict__ disp)
{
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * (NT / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (p >= N) return;
    const int dl = lane * C;
    const float *c = vol + (size_t)p * D + dl;
    float v[C];
    if (FULL) {
        const vecf<C> x = *reinterpret_cast<const vecf<C> *>(c);
#pragma unroll
        for (int k = 0; k < C; k++) v[k] = x.v[k];
    } else {
#pragma unroll
        for (int k = 0; k < C; k++) v[k] = (dl + k < D) ? c[k] : INFINITY;
    }
    const int wd = wave_wta<C, FULL>(v, dl, D);
    if (lane == 0) disp[p] = (float)wd;
}

}  // namespace

thread_local int g_smt_last_hip = 0;

struct smt_adcensus {
    int device;          // HIP device the handle lives on
    int H, W, D;
    float sigmaC, sigmaS;
    hipStream_t stream;
    float *vol[2];
    Tables T;            // table set used by the pair being issued (= TS[n_pairs & 1])
    Tables TS[2];        // two sets: the tables of pair n+1 are built while pair n's cost kernel runs
    hipStream_t prep_stream;                 // internal, non-blocking
    hipEvent_t in_ready, prep_done[2], cost_done[2];
    long n_pairs;        // pairs issued on this handle
    hipEvent_t *ev;      // SMT_TIMING_SLOTS * 4, lazily created
    bool timing;
    bool force_generic;  // test hook: route D%64==0 through the generic kernel too
    long n_timed;        // pairs recorded since timing was (re-)enabled
    long n_seen;         // pairs processed since timing was (re-)enabled
    int timing_stride;   // every timing_stride-th pair is recorded
    bool *ev_merged;     // slot recorded 3 events (tables end == cost start, one stream)
    bool plain_stores;   // both-views cost kernel with ordinary instead of streaming stores (chosen at create)
    float store_mode_ms[2];   // calibration: kernel ms with streaming / plain stores (0: not calibrated)
    int place_tries;     // candidate volume pairs tried by place_volumes
    float place_ms;      // store-only time of the pair that was kept (0: no search)
};

SMT_API const char *smt_strerror(int s)
{
    switch (s) {
    case SMT_OK: return "ok";
    case SMT_ERR_ARG: return "invalid argument";
    case SMT_ERR_HIP: return "HIP runtime error";
    case SMT_ERR_ALLOC: return "device allocation failed";
    case SMT_ERR_DOMAIN: return "image values outside the integer 0..255 domain";
    case SMT_ERR_REF_UB: return "reference behaviour undefined for these inputs";
    case SMT_ERR_STATE: return "call order violated";
    default: return "unknown status";
    }
}
SMT_API int smt_version(void) { return SMT_VERSION; }
SMT_API int smt_last_hip_error(void) { return g_smt_last_hip; }
SMT_API int smt_device_count(int *n)
{
    if (!n) return SMT_ERR_ARG;
    SMT_HIP(hipGetDeviceCount(n));
    return SMT_OK;
}
SMT_API int smt_set_device(int d) { SMT_HIP(hipSetDevice(d)); return SMT_OK; }
SMT_API int smt_malloc(void **p, size_t n)
{
    if (!p) return SMT_ERR_ARG;
    hipError_t e = hipMalloc(p, n ? n : 1);
    if (e != hipSuccess) { g_smt_last_hip = (int)e; return SMT_ERR_ALLOC; }
    return SMT_OK;
}
SMT_API int smt_free(void *p) { SMT_HIP(hipFree(p)); return SMT_OK; }
SMT_API int smt_memcpy_h2d(void *d, const void *s, size_t n, void *st)
{
    SMT_HIP(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, smt_stream(st)));
    return SMT_OK;
}
SMT_API int smt_memcpy_d2h(void *d, const void *s, size_t n, void *st)
{
    SMT_HIP(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, smt_stream(st)));
    return SMT_OK;
}
SMT_API int smt_memset(void *d, int b, size_t n, void *st)
{
    SMT_HIP(hipMemsetAsync(d, b, n, smt_stream(st)));
    return SMT_OK;
}
SMT_API int smt_stream_create(void **s)
{
    if (!s) return SMT_ERR_ARG;
    hipStream_t st;
    SMT_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *s = (void *)st;
    return SMT_OK;
}
SMT_API int smt_stream_destroy(void *s) { SMT_HIP(hipStreamDestroy(smt_stream(s))); return SMT_OK; }
SMT_API int smt_stream_sync(void *s) { SMT_HIP(hipStreamSynchronize(smt_stream(s))); return SMT_OK; }

template <int C>
static void launch_store_only(smt_adcensus *h, float *v0, float *v1, hipStream_t st)
{
    const int nbx = (h->W + FTJ - 1) / FTJ;
    const unsigned nblk = (unsigned)(((long)nbx * h->H * 2 + 7) / 8 * 8);
    hipLaunchKernelGGL((k_store_only2<C>), dim3(nblk), dim3(NT), 0, st, h->H, h->W, v0, v1, nbx);
}
static void store_only(smt_adcensus *h, float *v0, float *v1, hipStream_t st)
{
    switch (h->D / 64) {
    case 1: launch_store_only<1>(h, v0, v1, st); break;
    case 2: launch_store_only<2>(h, v0, v1, st); break;
    case 3: launch_store_only<3>(h, v0, v1, st); break;
    default: launch_store_only<4>(h, v0, v1, st); break;
    }
}

// Placement-aware allocation of the two cost volumes.  The cost kernel is a pure store stream (8 XCDs,
// each writing its own contiguous eighth of the two volumes), and what that stream reaches depends on
// WHICH physical pages hipMalloc handed out: on one MI355X, in one process, at a constant 2.37 GHz shader
// clock, the store-only twin of the kernel runs at either ~7.0 TB/s or ~5.9 TB/s on successive
// allocations of the same size, and the same virtual range re-allocated later lands in the other mode
// (tools/alloc_probe.hip).  Nothing in the kernel can change that afterwards, so
// Initialize takes up to PLACE_TRIES candidate pairs, times the store-only twin on each (a few launches,
// ~5 ms per candidate at 1080p x 192), keeps the fastest and frees the rest.  Rejected candidates stay
// allocated until the end so that every try gets different pages.  SMT_PLACEMENT=0 in the environment
// turns the search off (first allocation is used).
static int place_volumes(smt_adcensus *h, bool allow_search)
{
    const size_t V = (size_t)h->H * h->W * h->D;
    constexpr int PLACE_TRIES = 6;
    const char *env = getenv("SMT_PLACEMENT");
    const bool search = allow_search && !(env && env[0] == '0') && h->D % 64 == 0 && h->D <= 256 && V >= ((size_t)1 << 22);
    float *cand[PLACE_TRIES][2] = {};
    float ms[PLACE_TRIES];
    int n = 0, best = -1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (search && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) return SMT_ERR_HIP;
    // a candidate at this rate is in the fast mode (7.0 TB/s measured at 1080p x 192): stop looking
    const double good_ms = 2.0 * V * 4 / 6.75e12 * 1e3;
    for (; n < (search ? PLACE_TRIES : 1); n++) {
        if (smt_malloc((void **)&cand[n][0], V * 4) != SMT_OK) break;
        if (smt_malloc((void **)&cand[n][1], V * 4) != SMT_OK) { (void)hipFree(cand[n][0]); cand[n][0] = nullptr; break; }
        if (!search) { best = n; n++; break; }
        const int reps = 8;
        for (int k = 0; k < 3; k++) store_only(h, cand[n][0], cand[n][1], nullptr);
        (void)hipEventRecord(e0, nullptr);
        for (int k = 0; k < reps; k++) store_only(h, cand[n][0], cand[n][1], nullptr);
        (void)hipEventRecord(e1, nullptr);
        float t = 0;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess) t = 1e30f;
        ms[n] = t / reps;
        if (best < 0 || ms[n] < ms[best]) best = n;
        if (ms[n] <= good_ms) { n++; break; }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (int k = 0; k < n; k++)
        if (k != best) { (void)hipFree(cand[k][0]); (void)hipFree(cand[k][1]); }
    if (best < 0) return SMT_ERR_ALLOC;
    h->vol[0] = cand[best][0]; h->vol[1] = cand[best][1];
    h->place_tries = n;
    h->place_ms = search ? ms[best] : 0.0f;
    return SMT_OK;
}

template <int C, bool FULL>
static void launch_fast(smt_adcensus *h, int views, float *dL, float *dR, const float *nL = nullptr, const float *nR = nullptr);

// Streaming (non-temporal) or ordinary stores for the volumes?  Streaming stores were worth 8-19 % on the
// devices of round 1; on another device the store-only twin is 5 % FASTER with ordinary stores (DESIGN.md
// section 4).  Like the placement this is a property of the device, so Initialize times the real both-views
// kernel a few launches each way on the handle's own (zeroed) tables and volumes and keeps ordinary stores
// only when they win by more than 2 %.  SMT_STORE_MODE=nt / plain in the environment fixes the choice.
static void calibrate_store_mode(smt_adcensus *h, bool allow)
{
    h->plain_stores = false; h->store_mode_ms[0] = h->store_mode_ms[1] = 0.0f;
    const char *env = getenv("SMT_STORE_MODE");
    if (env && env[0] == 'p') { h->plain_stores = true; return; }
    if (env && env[0] == 'n') return;
    if (!allow) return;
    const size_t V = (size_t)h->H * h->W * h->D;
    if (h->D % 64 != 0 || V < ((size_t)1 << 22)) return;
    const size_t N = (size_t)h->H * h->W;
    // zeroed tables are valid inputs (census 0, bytes 0); the speed of a store-bound kernel does not depend on them
    for (int t = 0; t < 2; t++) {
        for (int v = 0; v < 2; v++) {
            if (hipMemset(h->TS[t].cenA[v], 0, N * 8) != hipSuccess || hipMemset(h->TS[t].u8[v], 0, N) != hipSuccess ||
                hipMemset(h->TS[t].cenX[v], 0, (size_t)h->H * h->TS[t].WX * 8) != hipSuccess) return;
        }
        if (hipMemset(h->TS[t].mask, 0, N * 8) != hipSuccess) return;
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return;
    const hipStream_t keep = h->stream;
    h->stream = nullptr;
    float best[2] = {1e30f, 1e30f};
    for (int round = 0; round < 2; round++)                    // interleaved: nt, plain, nt, plain
        for (int mode = 0; mode < 2; mode++) {
            h->plain_stores = mode == 1;
            const int reps = 4;
            auto launch = [&]() {
                switch (h->D / 64) {
                case 1: launch_fast<1, true>(h, SMT_VIEW_BOTH, nullptr, nullptr); break;
                case 2: launch_fast<2, true>(h, SMT_VIEW_BOTH, nullptr, nullptr); break;
                case 3: launch_fast<3, true>(h, SMT_VIEW_BOTH, nullptr, nullptr); break;
                default: launch_fast<4, true>(h, SMT_VIEW_BOTH, nullptr, nullptr); break;
                }
            };
            launch();
            (void)hipEventRecord(e0, nullptr);
            for (int k = 0; k < reps; k++) launch();
            (void)hipEventRecord(e1, nullptr);
            float t = 1e30f;
            if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&t, e0, e1) == hipSuccess) t /= reps;
            if (t < best[mode]) best[mode] = t;
        }
    h->stream = keep;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    h->store_mode_ms[0] = best[0]; h->store_mode_ms[1] = best[1];
    h->plain_stores = best[1] < 0.98f * best[0];
    // Initialize's contract: the volumes read as zeros until the first Compute* (AD-Census.h:341-342)
    (void)hipMemset(h->vol[0], 0, V * 4); (void)hipMemset(h->vol[1], 0, V * 4);
    (void)hipMemset(h->TS[0].flag, 0, 4);
}

SMT_API int smt_adcensus_store_mode(smt_adcensus *h, int *plain, float *nt_ms, float *plain_ms)
{
    if (!h) return SMT_ERR_ARG;
    if (plain) *plain = h->plain_stores ? 1 : 0;
    if (nt_ms) *nt_ms = h->store_mode_ms[0];
    if (plain_ms) *plain_ms = h->store_mode_ms[1];
    return SMT_OK;
}

SMT_API int smt_adcensus_placement(smt_adcensus *h, int *tries, float *store_only_ms)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    if (tries) *tries = h->place_tries;
    if (store_only_ms) *store_only_ms = h->place_ms;
    return SMT_OK;
}

static int adcensus_create(int H, int W, int D, float sigmaC, float sigmaS, unsigned flags, smt_adcensus **out)
{
    if (!out || H <= 0 || W <= 0 || D <= 0 || D > SMT_MAX_DISPARITY || !(sigmaC > 0.0f) || !(sigmaS > 0.0f))
        return SMT_ERR_ARG;
    smt_adcensus *h = new (std::nothrow) smt_adcensus();
    if (!h) return SMT_ERR_ALLOC;
    h->device = smt_current_device();
    h->H = H; h->W = W; h->D = D; h->sigmaC = sigmaC; h->sigmaS = sigmaS;
    h->stream = nullptr; h->timing = false; h->force_generic = false; h->ev = nullptr; h->n_timed = 0; h->n_seen = 0; h->timing_stride = 1; h->ev_merged = nullptr;
    h->n_pairs = 0;
    const size_t N = (size_t)H * W, V = N * D;
    const int WX = W + 4;
    int rc = place_volumes(h, !(flags & SMT_ADCENSUS_NO_PLACEMENT_SEARCH));
    auto alloc = [&](void **p, size_t bytes) { if (rc == SMT_OK) rc = smt_malloc(p, bytes); };
    alloc((void **)&h->TS[0].lut, 320 * 4);
    alloc((void **)&h->TS[0].flag, 4);
    for (int t = 0; t < 2; t++) {
        h->TS[t].WX = WX;
        h->TS[t].stamp = nullptr;
        h->TS[t].lut = h->TS[0].lut;                     // shared
        h->TS[t].flag = h->TS[0].flag;
        for (int v = 0; v < 2; v++) {
            alloc((void **)&h->TS[t].cenA[v], N * 8);
            alloc((void **)&h->TS[t].cenX[v], (size_t)H * WX * 8);
            alloc((void **)&h->TS[t].u8[v], N);
        }
        alloc((void **)&h->TS[t].mask, N * 8);
    }
    h->T = h->TS[0];
    if (rc != SMT_OK) { smt_adcensus_destroy(h); return rc; }
    if (hipStreamCreateWithFlags(&h->prep_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->in_ready, hipEventDisableTiming) != hipSuccess) {
        smt_adcensus_destroy(h);
        return SMT_ERR_HIP;
    }
    for (int t = 0; t < 2; t++)
        if (hipEventCreateWithFlags(&h->prep_done[t], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->cost_done[t], hipEventDisableTiming) != hipSuccess) {
            smt_adcensus_destroy(h);
            return SMT_ERR_HIP;
        }
    // fusion tables, the reference's own float expression (AD-Census.h:287-288)
    float lut[320];
    for (int k = 0; k < 256; k++) lut[k] = 1.0f - expf(-((float)k / sigmaC));
    for (int k = 0; k < 64; k++) lut[256 + k] = 1.0f - expf(-((float)k / sigmaS));
    // the reference's `new float[size*dispRange]()` value-initialises the volumes (AD-Census.h:341-342):
    // GetPtrLeft/Right before the first Compute* reads zeros
    if (hipMemcpy(h->TS[0].lut, lut, sizeof(lut), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(h->vol[0], 0, V * 4) != hipSuccess || hipMemset(h->vol[1], 0, V * 4) != hipSuccess ||
        hipMemset(h->TS[0].flag, 0, 4) != hipSuccess) {
        smt_adcensus_destroy(h);
        return SMT_ERR_HIP;
    }
    calibrate_store_mode(h, !(flags & SMT_ADCENSUS_NO_STORE_CALIBRATION));
    *out = h;
    return SMT_OK;
}

SMT_API int smt_adcensus_create(int H, int W, int D, float sigmaC, float sigmaS, smt_adcensus **out)
{
    return adcensus_create(H, W, D, sigmaC, sigmaS, 0u, out);
}

SMT_API int smt_adcensus_create_ex(int device, int H, int W, int D, float sigmaC, float sigmaS, unsigned flags, smt_adcensus **out)
{
    if (flags & ~(unsigned)(SMT_ADCENSUS_NO_PLACEMENT_SEARCH | SMT_ADCENSUS_NO_STORE_CALIBRATION)) return SMT_ERR_ARG;
    if (device < 0) return adcensus_create(H, W, D, sigmaC, sigmaS, flags, out);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device >= n) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(device);
    return adcensus_create(H, W, D, sigmaC, sigmaS, flags, out);
}

SMT_API int smt_adcensus_create_on(int device, int H, int W, int D, float sigmaC, float sigmaS, smt_adcensus **out)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(device);
    return smt_adcensus_create(H, W, D, sigmaC, sigmaS, out);
}

SMT_API int smt_adcensus_destroy(smt_adcensus *h)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    if (h->prep_stream) { (void)hipStreamSynchronize(h->prep_stream); (void)hipStreamDestroy(h->prep_stream); }
    if (h->in_ready) (void)hipEventDestroy(h->in_ready);
    for (int t = 0; t < 2; t++) {
        if (h->prep_done[t]) (void)hipEventDestroy(h->prep_done[t]);
        if (h->cost_done[t]) (void)hipEventDestroy(h->cost_done[t]);
    }
    (void)hipFree(h->vol[0]); (void)hipFree(h->vol[1]);
    for (int t = 0; t < 2; t++) {
        for (int v = 0; v < 2; v++) {
            (void)hipFree(h->TS[t].cenA[v]); (void)hipFree(h->TS[t].cenX[v]); (void)hipFree(h->TS[t].u8[v]);
        }
        (void)hipFree(h->TS[t].mask);
    }
    (void)hipFree(h->TS[0].lut); (void)hipFree(h->TS[0].flag);
    if (h->ev) {
        for (int k = 0; k < SMT_TIMING_SLOTS * 4; k++) (void)hipEventDestroy(h->ev[k]);
        delete[] h->ev_merged;
        delete[] h->ev;
    }
    delete h;
    return SMT_OK;
}

SMT_API int smt_adcensus_set_stream(smt_adcensus *h, void *s)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    h->stream = smt_stream(s);
    return SMT_OK;
}

template <int C, bool FULL>
static void launch_cost(smt_adcensus *h, int view0, int nviews, float *d0, float *d1)
{
    const int D = h->D;
    dim3 grid((h->W + TJ - 1) / TJ, h->H, nviews);
    const int NX = TJ + D;
    size_t shm = (size_t)NX * 8 + TJ * 16 + 320 * 4 + NX + TJ;
    shm = (shm + 15) & ~(size_t)15;
    hipLaunchKernelGGL((k_cost<C, FULL>), grid, dim3(NT), shm, h->stream, h->H, h->W, D, h->T,
                       view0, h->vol[0], h->vol[1], d0, d1);
}

template <int C, bool FULL>
static void launch_fast(smt_adcensus *h, int views, float *dL, float *dR, const float *nL, const float *nR)
{
    const int nbx = (h->W + FTJ - 1) / FTJ;
    auto blocks = [&](int nviews) { return dim3((unsigned)(((long)nbx * h->H * nviews + 7) / 8 * 8)); };
    if (views == SMT_VIEW_BOTH && nL) {
        // this pair's cost workgroups + the next pair's table workgroups (into the other table set) in one launch
        const int ncost = (int)blocks(2).x;
        const int ptx = (h->W + PTW - 1) / PTW, pty = (h->H + PTH - 1) / PTH;
        const int eb = (h->H + PNT / 16 - 1) / (PNT / 16);
        const int nprep = ptx * (pty + (eb + ptx - 1) / ptx);
        const unsigned grid = 8u * (unsigned)fused_grid(ncost, nprep).groups;
        const Tables &Tn = h->TS[(h->n_pairs + 1) & 1];
        if (h->plain_stores)
            hipLaunchKernelGGL((k_cost_fast2p<C, FULL, false>), dim3(grid), dim3(NT), 0, h->stream, h->H, h->W, h->D, h->T,
                               h->vol[0], h->vol[1], dL, dR, nbx, ncost, nprep, nL, nR, Tn, ptx);
        else
            hipLaunchKernelGGL((k_cost_fast2p<C, FULL, true>), dim3(grid), dim3(NT), 0, h->stream, h->H, h->W, h->D, h->T,
                               h->vol[0], h->vol[1], dL, dR, nbx, ncost, nprep, nL, nR, Tn, ptx);
        return;
    }
    if (views == SMT_VIEW_BOTH) {
        if (h->plain_stores)
            hipLaunchKernelGGL((k_cost_fast2<C, FULL, false>), blocks(2), dim3(NT), 0, h->stream, h->H, h->W, h->D, h->T, h->vol[0],
                               h->vol[1], dL, dR, nbx);
        else
            hipLaunchKernelGGL((k_cost_fast2<C, FULL, true>), blocks(2), dim3(NT), 0, h->stream, h->H, h->W, h->D, h->T, h->vol[0],
                               h->vol[1], dL, dR, nbx);
        return;
    }
    if (views & SMT_VIEW_LEFT)
        hipLaunchKernelGGL((k_cost_fast<C, 0, FULL>), blocks(1), dim3(NT), 0, h->stream, h->H, h->W, h->D, h->T,
                           h->vol[0], dL, nbx);
    if (views & SMT_VIEW_RIGHT)
        hipLaunchKernelGGL((k_cost_fast<C, 1, FULL>), blocks(1), dim3(NT), 0, h->stream, h->H, h->W, h->D, h->T,
                           h->vol[1], dR, nbx);
}

// One pair into table set (n & 1).  Three schedules for the pairs of a batch:
//   fused (default, both views through the register-window kernel): the table workgroups of pair n + 1 ride in the
//     grid of pair n's cost kernel (k_cost_fast2p) -- one stream, one launch per pair, no events; `prepped` says this
//     pair's tables were built that way, `nL / nR` are the next pair's images (null for the last pair of a batch);
//   overlap (SMT_OVERLAP=1, and the schedule of single views / D > 256): the table kernels run on the handle's internal
//     stream beside the previous pair's cost kernel on the caller's stream; event edges:
//       cost_done[n-2] --> prep stream               (table set n&1 is free again)
//       prep_done[n]   --> caller stream --> cost kernel(s) --> cost_done[n]
//     only for pairs b >= 1 of a batch (their inputs were already ordered behind the caller's stream by pair 0);
//   in order on the caller's stream (single pairs, the first pair of a batch, SMT_OVERLAP=0).
// Measured on MI355X the overlap gives +18 % at 1242x375 D=256 and +2-3.5 % at 1920x1080 D=192 over the in-order form
// (the table kernels then take 0.09 instead of 0.04 ms, hidden behind a cost kernel that gets 1 % slower); the fused
// form against the overlap: DESIGN.md section 4.
static bool fast_both_views(const smt_adcensus *h, int views)
{
    return views == SMT_VIEW_BOTH && !h->force_generic && (h->D + 63) / 64 <= 4;
}

static int adcensus_pair(smt_adcensus *h, const float *L, const float *R, int views, float *dL,
                         float *dR, bool overlap, bool prepped = false, const float *nL = nullptr,
                         const float *nR = nullptr)
{
    const int H = h->H, W = h->W, D = h->D;
    const int set = (int)(h->n_pairs & 1);
    h->T = h->TS[set];
    const bool timed = h->timing && (h->n_seen++ % h->timing_stride) == 0;
    const long slot = h->n_timed % SMT_TIMING_SLOTS;
    hipEvent_t *ev = timed ? h->ev + 4 * slot : nullptr;
    // overlap == false: everything in order on the caller's stream (single pairs, first pair of a batch)
    hipStream_t ps = overlap ? h->prep_stream : h->stream;
    if (overlap && h->n_pairs >= 2) SMT_HIP(hipStreamWaitEvent(ps, h->cost_done[set], 0));
    if (timed) (void)hipEventRecord(ev[0], ps);
    if (!prepped) {
        // tile workgroups + the workgroups of the 13 border columns (16 rows each) in one launch
        const int tx = (W + PTW - 1) / PTW, ty = (H + PTH - 1) / PTH;
        const int eb = (H + PNT / 16 - 1) / (PNT / 16);
        hipLaunchKernelGGL(k_prep, dim3(tx, ty + (eb + tx - 1) / tx), dim3(PNT), 0, ps, L, R, H, W, h->T);
    }
    if (timed) (void)hipEventRecord(ev[1], ps);
    if (overlap) {
        SMT_HIP(hipEventRecord(h->prep_done[set], ps));
        SMT_HIP(hipStreamWaitEvent(h->stream, h->prep_done[set], 0));
    }
    // one stream: the event after the table kernels is also the cost kernel's start
    if (timed) { h->ev_merged[slot] = !overlap; if (overlap) (void)hipEventRecord(ev[2], h->stream); }
    const int view0 = (views & SMT_VIEW_LEFT) ? 0 : 1;
    const int nviews = (views == SMT_VIEW_BOTH) ? 2 : 1;
    const int C = (D + 63) / 64;
    const bool full = (D % 64) == 0;
    if (h->force_generic || C > 4) {
        // the table-lookup kernel of the first version: kept as an independent formulation, and the kernel for
        // 256 < D <= 512 (a lane then owns 5..8 consecutive hypotheses; the register-window kernel stops at 4)
        switch (C) {
        case 1: launch_cost<1, false>(h, view0, nviews, dL, dR); break;
        case 2: launch_cost<2, false>(h, view0, nviews, dL, dR); break;
        case 3: launch_cost<3, false>(h, view0, nviews, dL, dR); break;
        case 4: launch_cost<4, false>(h, view0, nviews, dL, dR); break;
        case 5: launch_cost<5, false>(h, view0, nviews, dL, dR); break;
        case 6: launch_cost<6, false>(h, view0, nviews, dL, dR); break;
        case 7: launch_cost<7, false>(h, view0, nviews, dL, dR); break;
        default: launch_cost<8, false>(h, view0, nviews, dL, dR); break;
        }
    } else {
        switch (C * 2 + (full ? 1 : 0)) {
        case 3: launch_fast<1, true>(h, views, dL, dR, nL, nR); break;
        case 5: launch_fast<2, true>(h, views, dL, dR, nL, nR); break;
        case 7: launch_fast<3, true>(h, views, dL, dR, nL, nR); break;
        case 9: launch_fast<4, true>(h, views, dL, dR, nL, nR); break;
        case 2: launch_fast<1, false>(h, views, dL, dR, nL, nR); break;
        case 4: launch_fast<2, false>(h, views, dL, dR, nL, nR); break;
        case 6: launch_fast<3, false>(h, views, dL, dR, nL, nR); break;
        case 8: launch_fast<4, false>(h, views, dL, dR, nL, nR); break;
        default: return SMT_ERR_ARG;
        }
    }
    if (timed) { (void)hipEventRecord(ev[3], h->stream); h->n_timed++; }
    // the two-stream schedule waits on this before it reuses the table set; the fused one is ordered by its stream
    if (!prepped && !nL) SMT_HIP(hipEventRecord(h->cost_done[set], h->stream));
    h->n_pairs++;
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

SMT_API int smt_adcensus_compute(smt_adcensus *h, const float *L, const float *R, int views,
                                 float *dispL, float *dispR)
{
    if (!h || !L || !R || views < 1 || views > 3) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    return adcensus_pair(h, L, R, views, dispL, dispR, false);
}

// schedule of the pairs b >= 1 of a batch: 0 in order, 1 two streams, 2 fused launches (adcensus_pair)
static int batch_schedule(const smt_adcensus *h, int views)
{
    const char *env = getenv("SMT_OVERLAP");             // tuning hook: 0 / 1 / 2 forces the choice
    int want = 2;
    if (env && env[0] >= '0' && env[0] <= '2' && env[1] == 0) want = env[0] - '0';
    if (want == 2 && !fast_both_views(h, views)) want = 1;
    return want;
}

SMT_API int smt_adcensus_compute_batch(smt_adcensus *h, const float *L, const float *R, int pairs,
                                       int views, float *dispL, float *dispR)
{
    if (!h || !L || !R || pairs <= 0 || views < 1 || views > 3) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    const size_t N = (size_t)h->H * h->W;
    const int sched = pairs > 1 ? batch_schedule(h, views) : 0;
    if (sched == 1) {
        // order the internal stream behind whatever produced the [pairs][H][W] inputs
        SMT_HIP(hipEventRecord(h->in_ready, h->stream));
        SMT_HIP(hipStreamWaitEvent(h->prep_stream, h->in_ready, 0));
    }
    for (int b = 0; b < pairs; b++) {
        const bool more = sched == 2 && b + 1 < pairs;
        int rc = adcensus_pair(h, L + b * N, R + b * N, views, dispL ? dispL + b * N : nullptr,
                               dispR ? dispR + b * N : nullptr, sched == 1 && b > 0, sched == 2 && b > 0,
                               more ? L + (b + 1) * N : nullptr, more ? R + (b + 1) * N : nullptr);
        if (rc != SMT_OK) return rc;
    }
    return SMT_OK;
}

SMT_API int smt_adcensus_volume(smt_adcensus *h, int view, float **vol)
{
    if (!h || !vol || (view != SMT_VIEW_LEFT && view != SMT_VIEW_RIGHT)) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    *vol = h->vol[view == SMT_VIEW_LEFT ? 0 : 1];
    return SMT_OK;
}

SMT_API int smt_adcensus_status(smt_adcensus *h)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    int f = 0;
    SMT_HIP(hipMemcpyAsync(&f, h->T.flag, 4, hipMemcpyDeviceToHost, h->stream));
    SMT_HIP(hipMemsetAsync(h->T.flag, 0, 4, h->stream));            // read-and-clear
    SMT_HIP(hipStreamSynchronize(h->stream));
    return f ? SMT_ERR_DOMAIN : SMT_OK;
}

// Host-side check of the fused launch's workgroup arithmetic (fused_grid / fused_decode, the functions the kernel runs):
// every cost group and every table group of a launch with `ncost` cost and `nprep` table workgroups is reached exactly
// once.  Needs no GPU.
SMT_API int smt_adcensus_selftest_fused_grid(int ncost, int nprep)
{
    if (ncost <= 0 || (ncost & 7) || nprep <= 0) return SMT_ERR_ARG;
    const FusedGrid f = fused_grid(ncost, nprep);
    if (f.groups < f.cgroups + f.pgroups) return SMT_ERR_STATE;
    unsigned char *seen = new (std::nothrow) unsigned char[(size_t)2 * f.groups]();
    if (!seen) return SMT_ERR_ALLOC;
    int rc = SMT_OK, ntab = 0;
    for (int g = 0; g < f.groups && rc == SMT_OK; g++) {
        int idx = -1;
        const bool table = fused_decode(f, g, idx);
        if (idx < 0 || idx >= f.groups || (!table && idx >= f.cgroups)) { rc = SMT_ERR_STATE; break; }
        unsigned char &cell = seen[(table ? f.groups : 0) + idx];
        if (cell) rc = SMT_ERR_STATE;
        cell = 1;
        ntab += table;
    }
    for (int c = 0; c < f.cgroups && rc == SMT_OK; c++) if (!seen[c]) rc = SMT_ERR_STATE;              // every cost group
    for (int t = 0; t < ntab && rc == SMT_OK; t++) if (!seen[f.groups + t]) rc = SMT_ERR_STATE;         // table groups 0 .. ntab - 1
    if (rc == SMT_OK && ntab < f.pgroups) rc = SMT_ERR_STATE;
    delete[] seen;
    return rc;
}

SMT_API int smt_adcensus_force_generic(smt_adcensus *h, int on)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    h->force_generic = on != 0;
    return SMT_OK;
}

SMT_API int smt_adcensus_timing(smt_adcensus *h, int enable)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    if (enable && !h->ev) {
        h->ev = new (std::nothrow) hipEvent_t[SMT_TIMING_SLOTS * 4];
        h->ev_merged = new (std::nothrow) bool[SMT_TIMING_SLOTS]();
        if (!h->ev || !h->ev_merged) return SMT_ERR_ALLOC;
        for (int k = 0; k < SMT_TIMING_SLOTS * 4; k++) SMT_HIP(hipEventCreate(&h->ev[k]));
    }
    if (enable < 0) return SMT_ERR_ARG;
    h->timing = enable != 0;
    h->timing_stride = enable > 0 ? enable : 1;
    h->n_timed = 0;
    h->n_seen = 0;
    return SMT_OK;
}

SMT_API int smt_adcensus_kernel_times(smt_adcensus *h, float *prep_ms, float *cost_ms, int capacity,
                                      int *count)
{
    if (!h || !count || capacity < 0) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    if (!h->ev) return SMT_ERR_STATE;
    long n = h->n_timed < SMT_TIMING_SLOTS ? h->n_timed : SMT_TIMING_SLOTS;
    if (n > capacity) n = capacity;
    const long first = h->n_timed - n;
    for (long k = 0; k < n; k++) {
        hipEvent_t *ev = h->ev + 4 * ((first + k) % SMT_TIMING_SLOTS);
        SMT_HIP(hipEventSynchronize(ev[3]));
        float a = 0, b = 0;
        SMT_HIP(hipEventElapsedTime(&a, ev[0], ev[1]));      // table kernels, internal stream
        const long slot = (first + k) % SMT_TIMING_SLOTS;
        SMT_HIP(hipEventElapsedTime(&b, ev[h->ev_merged[slot] ? 1 : 2], ev[3]));   // cost kernel(s), caller's stream
        if (prep_ms) prep_ms[k] = a;
        if (cost_ms) cost_ms[k] = b;
    }
    *count = (int)n;
    return SMT_OK;
}

// Measurement hook (bench.py): see include/smt.h.
namespace {
// everything smt_adcensus_diag_impl acquires, released on every exit path (its SMT_HIP early returns included)
struct DiagResources {
    hipEvent_t e[3] = {nullptr, nullptr, nullptr};
    unsigned long long *stamp = nullptr;
    unsigned long long *hs = nullptr;
    float *ratio = nullptr;
    ~DiagResources()
    {
        for (auto &x : e) if (x) (void)hipEventDestroy(x);
        if (stamp) (void)hipFree(stamp);
        delete[] hs;
        delete[] ratio;
    }
};
}  // namespace

static int smt_adcensus_diag_impl(smt_adcensus *h, int reps, float *sclk_mhz, float *cost_ms, float *store_only_ms, DiagResources &res)
{
    if (h->n_pairs == 0) return SMT_ERR_STATE;               // needs the tables of a computed pair
    const int D = h->D, C = D / 64;
    if (D % 64 != 0 || C < 1 || C > 4) return SMT_ERR_ARG;
    const int nbx = (h->W + FTJ - 1) / FTJ;
    const unsigned nblk = (unsigned)(((long)nbx * h->H * 2 + 7) / 8 * 8);
    hipEvent_t (&e)[3] = res.e;
    for (auto &x : e) SMT_HIP(hipEventCreate(&x));
    if (smt_malloc((void **)&res.stamp, (size_t)nblk * 32) != SMT_OK) return SMT_ERR_ALLOC;
    unsigned long long *stamp = res.stamp;
    SMT_HIP(hipMemsetAsync(stamp, 0, (size_t)nblk * 32, h->stream));
    Tables T = h->T;
    T.stamp = stamp;
    SMT_HIP(hipEventRecord(e[0], h->stream));
    for (int r = 0; r < reps; r++) {
        switch (C) {
        case 1: hipLaunchKernelGGL((k_store_only2<1>), dim3(nblk), dim3(NT), 0, h->stream, h->H, h->W, h->vol[0], h->vol[1], nbx); break;
        case 2: hipLaunchKernelGGL((k_store_only2<2>), dim3(nblk), dim3(NT), 0, h->stream, h->H, h->W, h->vol[0], h->vol[1], nbx); break;
        case 3: hipLaunchKernelGGL((k_store_only2<3>), dim3(nblk), dim3(NT), 0, h->stream, h->H, h->W, h->vol[0], h->vol[1], nbx); break;
        default: hipLaunchKernelGGL((k_store_only2<4>), dim3(nblk), dim3(NT), 0, h->stream, h->H, h->W, h->vol[0], h->vol[1], nbx); break;
        }
    }
    SMT_HIP(hipEventRecord(e[1], h->stream));
    // the real kernel with stamps, last, so that the volumes hold the pair's costs again afterwards
    for (int r = 0; r < reps; r++) {
        switch (C) {
        case 1: if (h->plain_stores) hipLaunchKernelGGL((k_cost_fast2<1, true, false>), dim3(nblk), dim3(NT), 0, h->stream, h->H, h->W, D, T, h->vol[0], h->vol[1], (float *)nullptr, (float *)nullptr, nbx);
            else hipLaunchKernelGGL((k_cost_fast2<1, true, true>), dim3(nblk), dim3(NT), 0, h->stream, h->H, h->W, D, T, h->vol[0], h->vol[1], (float *)nullptr, (float *)nullptr, nbx);
            break;
        case 2: if (h->plain_stores) hipLaunchKernelGGL((k_cost_fast2<2, true, false>), dim3(nblk), dim3(NT), 0, h->stream, h->H, h->W, D, T, h->vol[0], h->vol[1], (float *)nullptr, (float *)nullptr, nbx);
            else hipLaunchKernelGGL((k_cost_fast2<2, true, true>), dim3(nblk), dim3(NT), 0, h->stream, h->H, h->W, D, T, h->vol[0], h->vol[1], (float *)nullptr, (float *)nullptr, nbx);
            break;
        case 3: if (h->plain_stores) hipLaunchKernelGGL((k_cost_fast2<3, true, false>), dim3(nblk), dim3(NT), 0, h->stream, h->H, h->W, D, T, h->vol[0], h->vol[1], (float *)nullptr, (float *)nullptr, nbx);
            else hipLaunchKernelGGL((k_cost_fast2<3, true, true>), dim3(nblk), dim3(NT), 0, h->stream, h->H, h->W, D, T, h->vol[0], h->vol[1], (float *)nullptr, (float *)nullptr, nbx);
            break;
        default: if (h->plain_stores) hipLaunchKernelGGL((k_cost_fast2<4, true, false>), dim3(nblk), dim3(NT), 0, h->stream, h->H, h->W, D, T, h->vol[0], h->vol[1], (float *)nullptr, (float *)nullptr, nbx);
            else hipLaunchKernelGGL((k_cost_fast2<4, true, true>), dim3(nblk), dim3(NT), 0, h->stream, h->H, h->W, D, T, h->vol[0], h->vol[1], (float *)nullptr, (float *)nullptr, nbx);
            break;
        }
    }
    SMT_HIP(hipEventRecord(e[2], h->stream));
    SMT_LAUNCH_CHECK();
    SMT_HIP(hipEventSynchronize(e[2]));
    float a = 0, b = 0;
    SMT_HIP(hipEventElapsedTime(&a, e[0], e[1]));
    SMT_HIP(hipEventElapsedTime(&b, e[1], e[2]));
    if (store_only_ms) *store_only_ms = a / reps;
    if (cost_ms) *cost_ms = b / reps;
    if (sclk_mhz) {
        // stamps of the last launch: median over workgroups of d(s_memtime) / d(s_memrealtime) * 100 MHz
        unsigned long long *hs = res.hs = new (std::nothrow) unsigned long long[(size_t)nblk * 4];
        float *ratio = res.ratio = new (std::nothrow) float[nblk];
        if (!hs || !ratio) return SMT_ERR_ALLOC;
        SMT_HIP(hipMemcpy(hs, stamp, (size_t)nblk * 32, hipMemcpyDeviceToHost));
        size_t n = 0;
        for (unsigned k = 0; k < nblk; k++) {
            const unsigned long long dt = hs[4 * k + 2] - hs[4 * k], dr = hs[4 * k + 3] - hs[4 * k + 1];
            if (hs[4 * k + 3] != 0 && dr >= 200) ratio[n++] = (float)((double)dt / (double)dr * 100.0);
        }
        if (n) { std::nth_element(ratio, ratio + n / 2, ratio + n); *sclk_mhz = ratio[n / 2]; }
        else *sclk_mhz = 0.0f;
    }
    return SMT_OK;
}

SMT_API int smt_adcensus_diag(smt_adcensus *h, int reps, float *sclk_mhz, float *cost_ms, float *store_only_ms)
{
    if (!h || reps <= 0 || reps > 1000) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    DiagResources res;
    return smt_adcensus_diag_impl(h, reps, sclk_mhz, cost_ms, store_only_ms, res);
}

SMT_API int smt_wta(const float *vol, int H, int W, int D, float *disp, void *stream)
{
    if (!vol || !disp || H <= 0 || W <= 0 || D <= 0 || D > SMT_MAX_DISPARITY) return SMT_ERR_ARG;
    const int N = H * W;
    dim3 grid((N + 3) / 4);
    const int C = (D + 63) / 64;
    hipStream_t st = smt_stream(stream);
    const bool full = (D == 64 * C) && C <= 4;
#define SMT_WTA(CC)                                                                                  \
    do {                                                                                             \
        if (full) hipLaunchKernelGGL((k_wta<CC, true>), grid, dim3(NT), 0, st, vol, N, D, disp);     \
        else hipLaunchKernelGGL((k_wta<CC, false>), grid, dim3(NT), 0, st, vol, N, D, disp);         \
    } while (0)
    switch (C) {
    case 1: SMT_WTA(1); break;
    case 2: SMT_WTA(2); break;
    case 3: SMT_WTA(3); break;
    case 4: SMT_WTA(4); break;
    case 5: hipLaunchKernelGGL((k_wta<5, false>), grid, dim3(NT), 0, st, vol, N, D, disp); break;
    case 6: hipLaunchKernelGGL((k_wta<6, false>), grid, dim3(NT), 0, st, vol, N, D, disp); break;
    case 7: hipLaunchKernelGGL((k_wta<7, false>), grid, dim3(NT), 0, st, vol, N, D, disp); break;
    default: hipLaunchKernelGGL((k_wta<8, false>), grid, dim3(NT), 0, st, vol, N, D, disp); break;
    }
#undef SMT_WTA
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}
