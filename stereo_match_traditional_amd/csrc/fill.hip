// Hole filling -- replaces FillTheHole (AD-CensusV1/PostProcessing.h:156-248; CBLSM/PostProcessing.h
// holds the same text).
//
// The reference walks three target lists (occlusions, mismatches, then every entry that still
// equals 65535).  For each target it casts 8 rays, takes the first entry != 65535 on each, sorts
// what it found and keeps the second smallest (pass 0) or the median (passes 1, 2); all writes
// of a pass happen after all its reads (:240-245), so a pass is data-parallel over its targets.
// Two pieces of sequential state are restated in closed form:
//   * `angle` switches from angle1 to angle2 at the first target whose first coordinate equals
//     height/2 and never switches back, across passes (:166, :195-197): the host finds that
//     index in each list; in pass 2 the list is in raster order, so the switch happens at the
//     first hole of line height/2, i.e. for every hole with y >= height/2 once that line has one;
//   * a pixel listed twice is written twice, the later entry wins (:240-245): the apply step
//     keeps, per pixel, the entry with the largest list index.
// The reference swaps the extents (`width = row`, `height = col`, :158-159); this is reproduced: the
// buffer is addressed as `col` lines of `row` entries.
//
// One group of 8 lanes per target, lane s = ray s; ray positions are float arithmetic
// (`y + m * sina` with int y, m and float sina, then lround, :206-207).  sin/cos of the 16 float
// angles come from the host libm (sinf/cosf: `sin(float)` is the float overload in C++), like the
// AD-Census LUTs.
#include "smt_common.h"
#include <cmath>
#include <climits>
#include <cstring>
#include <vector>

namespace {

constexpr int NT = 256;
constexpr float HOLE = 65535.0f;                      // 0xffff, :182, :212

struct FillCfg {
    int width, height, maxlen;
    float sn[2][8], cs[2][8];
};

__device__ __forceinline__ long lround_f(float v)     // lround(float): half away from zero
{
    const double d = (double)v;                        // d +- 0.5 is exact in double
    return (long)(d + (d >= 0.0 ? 0.5 : -0.5));
}

// first entry != 65535 along ray s of set `set` from (y, x); false if the ray leaves the buffer first
__device__ __forceinline__ bool ray(const float *__restrict__ disp, const FillCfg &c, int y, int x, int set, int s,
                                    float &val)
{
    const float sina = c.sn[set][s], cosa = c.cs[set][s];
    for (int m = 1; m < c.maxlen; m++) {
        const long yy = lround_f((float)y + (float)m * sina);
        const long xx = lround_f((float)x + (float)m * cosa);
        if (yy < 0 || yy >= c.height || xx < 0 || xx >= c.width) return false;
        const float d = disp[yy * c.width + xx];
        if (d != HOLE) { val = d; return true; }
    }
    return false;
}

// the 8 lanes of a group hold (found, val); returns the reference's pick: element `1 (or 0)` of
// the sorted finds for kind 0, element ng/2 for kind 1; 0.0f when nothing was found (:177, :218)
__device__ __forceinline__ float pick(bool found, float val, int s, int kind)
{
    int rank = 0, ng = 0;
    float vals[8];
    bool fs[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        vals[j] = __shfl(val, j, 8);
        fs[j] = __shfl((int)found, j, 8) != 0;
        if (fs[j]) {
            ng++;
            if (vals[j] < val || (vals[j] == val && j < s)) rank++;
        }
    }
    const int want = (kind == 0) ? (ng > 1 ? 1 : 0) : ng / 2;
    // exactly one finding lane has rank == want; everyone learns its value
    float out = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int rj = __shfl(rank, j, 8);
        if (fs[j] && rj == want) out = vals[j];
    }
    return ng == 0 ? 0.0f : out;
}

__global__ void __launch_bounds__(NT) k_fill_collect(const float *__restrict__ disp, FillCfg c,
                                                     const int *__restrict__ pairs, int n, int switch_from, int kind,
                                                     float *__restrict__ fill, int *__restrict__ winner)
{
    const long tid = (long)blockIdx.x * NT + threadIdx.x;
    const int t = (int)(tid >> 3), s = (int)(tid & 7);
    const bool live = t < n;                           // whole groups are live or not
    const int tt = live ? t : 0;
    const int y = pairs[2 * tt], x = pairs[2 * tt + 1];
    float val = 0.0f;
    const bool found = live && ray(disp, c, y, x, t >= switch_from ? 1 : 0, s, val);
    const float v = pick(found, val, s, kind);
    if (live && s == 0) {
        fill[t] = v;
        atomicMax(&winner[(long)y * c.width + x], t);
    }
}

__global__ void __launch_bounds__(NT) k_fill_apply(float *__restrict__ disp, int width, const int *__restrict__ pairs,
                                                   int n, const float *__restrict__ fill, const int *__restrict__ winner)
{
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= n) return;
    const long at = (long)pairs[2 * t] * width + pairs[2 * t + 1];
    if (winner[at] == t) disp[at] = fill[t];
}

// counts[0] = number of holes, counts[1] = 1 if line height/2 holds one
__global__ void __launch_bounds__(NT) k_fill_count(const float *__restrict__ disp, int width, int height, int *counts)
{
    const long n = (long)width * height;
    int mine = 0, mid = 0;
    for (long p = (long)blockIdx.x * NT + threadIdx.x; p < n; p += (long)gridDim.x * NT)
        if (disp[p] == HOLE) { mine++; if (p / width == height / 2) mid = 1; }
    __shared__ int s_cnt, s_mid;
    if (threadIdx.x == 0) { s_cnt = 0; s_mid = 0; }
    __syncthreads();
    if (mine) atomicAdd(&s_cnt, mine);
    if (mid) atomicOr(&s_mid, 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_cnt) atomicAdd(&counts[0], s_cnt);
        if (s_mid) atomicOr(&counts[1], 1);
    }
}

// pass 2: every hole, median rule; out = filled copy of disp
__global__ void __launch_bounds__(NT) k_fill_holes(const float *__restrict__ disp, FillCfg c, int switched, int midline,
                                                   float *__restrict__ out)
{
    const long tid = (long)blockIdx.x * NT + threadIdx.x;
    const long p = tid >> 3;
    const int s = (int)(tid & 7);
    const long n = (long)c.width * c.height;
    const bool live = p < n;
    const float d0 = live ? disp[p] : 0.0f;
    const bool hole = live && d0 == HOLE;
    // the shuffles in pick() need whole waves; a wave without holes has nothing to do
    if (!__any((int)hole)) {
        if (live && s == 0) out[p] = d0;
        return;
    }
    const int y = (int)(p / c.width), x = (int)(p - (long)y * c.width);
    const int set = (switched || (midline && y >= c.height / 2)) ? 1 : 0;
    float val = 0.0f;
    const bool found = hole && ray(disp, c, y, x, set, s, val);
    const float v = pick(found, val, s, 1);
    if (live && s == 0) out[p] = hole ? v : d0;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4) == hipSuccess ? SMT_OK : SMT_ERR_ALLOC; }
};

}  // namespace

SMT_API int smt_fill_the_hole(float *disp, int row, int col, int dispRange, const int *occ, int n_occ,
                              const int *mis, int n_mis, int *third, int *n_third, void *stream)
{
    if (!disp || row <= 0 || col <= 0 || dispRange < 0 || n_occ < 0 || n_mis < 0 || (n_occ && !occ) ||
        (n_mis && !mis))
        return SMT_ERR_ARG;
    hipStream_t st = smt_stream(stream);
    FillCfg c;
    c.width = row; c.height = col;                                           // :158-159
    c.maxlen = (int)(1.0 * dispRange);                                       // :168
    const float pi = 3.1415926f;
    const float angle1[8] = {pi, 3 * pi / 4, pi / 2, pi / 4, 0, 7 * pi / 4, 3 * pi / 2, 5 * pi / 4};
    const float angle2[8] = {pi, 5 * pi / 4, 3 * pi / 2, 7 * pi / 4, 0, pi / 4, pi / 2, 3 * pi / 4};
    for (int s = 0; s < 8; s++) {
        c.sn[0][s] = sinf(angle1[s]); c.cs[0][s] = cosf(angle1[s]);
        c.sn[1][s] = sinf(angle2[s]); c.cs[1][s] = cosf(angle2[s]);
    }
    const long n = (long)row * col;
    if (n_third) *n_third = -1;
    // a listed pair outside the buffer is an out-of-bounds write in the reference (:244)
    for (int k = 0; k < 2; k++) {
        const int *trg = k == 0 ? occ : mis;
        const int nt = k == 0 ? n_occ : n_mis;
        for (int t = 0; t < nt; t++) {
            const long at = (long)trg[2 * t] * c.width + trg[2 * t + 1];
            if (at < 0 || at >= n) return SMT_ERR_REF_UB;
        }
    }
    bool switched = false;
    DevBuf winner;
    for (int k = 0; k < 2; k++) {
        const int *trg = k == 0 ? occ : mis;
        const int nt = k == 0 ? n_occ : n_mis;
        if (nt == 0) continue;                                                // :174-176
        int switch_from = switched ? 0 : INT_MAX;
        if (!switched)
            for (int t = 0; t < nt; t++)
                if (trg[2 * t] == c.height / 2) { switch_from = t; switched = true; break; }
        DevBuf pairs, fill;
        if (pairs.alloc((size_t)nt * 8) != SMT_OK || fill.alloc((size_t)nt * 4) != SMT_OK) return SMT_ERR_ALLOC;
        if (!winner.p && winner.alloc((size_t)n * 4) != SMT_OK) return SMT_ERR_ALLOC;
        SMT_HIP(hipMemcpyAsync(pairs.p, trg, (size_t)nt * 8, hipMemcpyHostToDevice, st));
        SMT_HIP(hipMemsetAsync(winner.p, 0xff, (size_t)n * 4, st));         // -1
        const long threads = (long)nt * 8;
        hipLaunchKernelGGL(k_fill_collect, dim3((unsigned)((threads + NT - 1) / NT)), dim3(NT), 0, st, disp, c,
                           (const int *)pairs.p, nt, switch_from, k, (float *)fill.p, (int *)winner.p);
        hipLaunchKernelGGL(k_fill_apply, dim3((unsigned)((nt + NT - 1) / NT)), dim3(NT), 0, st, disp, c.width,
                           (const int *)pairs.p, nt, (const float *)fill.p, (const int *)winner.p);
        SMT_LAUNCH_CHECK();
        SMT_HIP(hipStreamSynchronize(st));                                   // the list buffers go out of scope
    }
    if (n_mis == 0) return SMT_OK;                                            // the third pass tests the mismatch list, :174
    // third pass: every entry that still equals 65535 (:179-187)
    DevBuf counts;
    if (counts.alloc(8) != SMT_OK) return SMT_ERR_ALLOC;
    SMT_HIP(hipMemsetAsync(counts.p, 0, 8, st));
    const int blocks = (int)((n + NT - 1) / NT < 1024 ? (n + NT - 1) / NT : 1024);
    hipLaunchKernelGGL(k_fill_count, dim3(blocks), dim3(NT), 0, st, disp, c.width, c.height, (int *)counts.p);
    SMT_LAUNCH_CHECK();
    int hc[2] = {0, 0};
    SMT_HIP(hipMemcpyAsync(hc, counts.p, 8, hipMemcpyDeviceToHost, st));
    SMT_HIP(hipStreamSynchronize(st));
    if (third || n_third) {
        // the list the reference leaves in the caller's `mismatch` vector (:186), raster order
        if (third && hc[0] > 0) {
            std::vector<float> host((size_t)n);
            SMT_HIP(hipMemcpyAsync(host.data(), disp, (size_t)n * 4, hipMemcpyDeviceToHost, st));
            SMT_HIP(hipStreamSynchronize(st));
            int cnt = 0;
            for (int i = 0; i < c.height; i++)
                for (int j = 0; j < c.width; j++)
                    if (host[(size_t)i * c.width + j] == HOLE) { third[2 * cnt] = i; third[2 * cnt + 1] = j; cnt++; }
        }
        if (n_third) *n_third = hc[0];
    }
    // `fill_disps` keeps the size of the old mismatch list (:177 before :186): more holes than
    // that is an out-of-bounds write in the reference
    if (hc[0] > n_mis) return SMT_ERR_REF_UB;
    if (hc[0] == 0) return SMT_OK;
    DevBuf tmp;
    if (tmp.alloc((size_t)n * 4) != SMT_OK) return SMT_ERR_ALLOC;
    const long threads = n * 8;
    hipLaunchKernelGGL(k_fill_holes, dim3((unsigned)((threads + NT - 1) / NT)), dim3(NT), 0, st, disp, c,
                       switched ? 1 : 0, hc[1], (float *)tmp.p);
    SMT_LAUNCH_CHECK();
    SMT_HIP(hipMemcpyAsync(disp, tmp.p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    SMT_HIP(hipStreamSynchronize(st));
    return SMT_OK;
}
