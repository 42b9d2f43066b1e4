// Data formats either side of the path: what the reference's main()s do
// right before and right after the hot path, so a caller can stay on the device.
//   - BGR -> gray: cvtColor(CV_BGR2GRAY) (AD-CensusV1/main.cpp:19-20, CBLSM.cpp:21-22).  OpenCV 3.1.0
//     (not in this image; pinned by */*.vcxproj linker lines) computes 8-bit gray in fixed point:
//     (1868*B + 9617*G + 4899*R + (1 << 13)) >> 14   (RGB2Gray<uchar>, yuv_shift = 14).
//   - replicate padding: copyMakeBorder(BORDER_REPLICATE) (SADmain.cpp:47-48, ASWeight.cpp:54-57).
//   - uchar -> float staging (main.cpp:46-55).
//   - MedianFilter (AD-CensusV1/PostProcessing.h:314-344): sort the in-image part of the
//     wnd x wnd window, take element [n/2].
#include "smt_common.h"

namespace {

constexpr int NT = 256;

__global__ void __launch_bounds__(NT) k_bgr2gray(const uint8_t *__restrict__ bgr, int n, uint8_t *__restrict__ gray)
{
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= n) return;
    const int b = bgr[3 * p], g = bgr[3 * p + 1], r = bgr[3 * p + 2];
    gray[p] = (uint8_t)((1868 * b + 9617 * g + 4899 * r + (1 << 13)) >> 14);
}

__global__ void __launch_bounds__(NT) k_pad(const uint8_t *__restrict__ src, int H, int W, int pad,
                                            uint8_t *__restrict__ dst)
{
    const int Wp = W + 2 * pad, Hp = H + 2 * pad;
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= Hp * Wp) return;
    const int ip = p / Wp, jp = p - ip * Wp;
    int i = ip - pad, j = jp - pad;
    i = i < 0 ? 0 : (i > H - 1 ? H - 1 : i);
    j = j < 0 ? 0 : (j > W - 1 ? W - 1 : j);
    dst[p] = src[(size_t)i * W + j];
}

__global__ void __launch_bounds__(NT) k_u8_to_f32(const uint8_t *__restrict__ src, int n, float *__restrict__ dst)
{
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p < n) dst[p] = (float)src[p];                                    // static_cast<float>(uchar), main.cpp:52-53
}

// one thread per pixel; window <= 7x7.  std::sort's result on floats (with +inf, no NaN) is the
// ascending order, so element [n/2] is order-independent of the sort algorithm.
__global__ void __launch_bounds__(NT) k_median(const float *__restrict__ in, float *__restrict__ out, int W, int H,
                                               int radius)
{
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= W * H) return;
    const int i = p / W, j = p - i * W;
    float v[49];
    int n = 0;
    for (int r = -radius; r <= radius; r++)
        for (int c = -radius; c <= radius; c++) {
            const int row = i + r, col = j + c;
            if (row >= 0 && row < H && col >= 0 && col < W) {
                // insertion into the sorted prefix
                const float x = in[(size_t)row * W + col];
                int k = n++;
                while (k > 0 && v[k - 1] > x) { v[k] = v[k - 1]; k--; }
                v[k] = x;
            }
        }
    out[p] = v[n / 2];
}

}  // namespace

SMT_API int smt_bgr2gray(const uint8_t *bgr, int H, int W, uint8_t *gray, void *stream)
{
    if (!bgr || !gray || H <= 0 || W <= 0) return SMT_ERR_ARG;
    const int n = H * W;
    hipLaunchKernelGGL(k_bgr2gray, dim3((n + NT - 1) / NT), dim3(NT), 0, smt_stream(stream), bgr, n, gray);
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

SMT_API int smt_pad_replicate(const uint8_t *src, int H, int W, int pad, uint8_t *dst, void *stream)
{
    if (!src || !dst || H <= 0 || W <= 0 || pad < 0) return SMT_ERR_ARG;
    const int n = (H + 2 * pad) * (W + 2 * pad);
    hipLaunchKernelGGL(k_pad, dim3((n + NT - 1) / NT), dim3(NT), 0, smt_stream(stream), src, H, W, pad, dst);
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

SMT_API int smt_u8_to_f32(const uint8_t *src, int H, int W, float *dst, void *stream)
{
    if (!src || !dst || H <= 0 || W <= 0) return SMT_ERR_ARG;
    const int n = H * W;
    hipLaunchKernelGGL(k_u8_to_f32, dim3((n + NT - 1) / NT), dim3(NT), 0, smt_stream(stream), src, n, dst);
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

// float64 sum of a float32 array (the checksum of the multi-GPU gather): per-wave DPP-free
// shuffle reduction + one atomic per workgroup.  *out_dev is zeroed on the stream first.
__global__ void __launch_bounds__(256) k_sum_f32(const float *__restrict__ x, size_t n, double *out)
{
    double acc = 0.0;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) acc += (double)x[k];
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s[0] + s[1] + s[2] + s[3]);
}

SMT_API int smt_sum_f32(const float *x, size_t n, double *out_dev, void *stream)
{
    if (!x || !out_dev) return SMT_ERR_ARG;
    SMT_HIP(hipMemsetAsync(out_dev, 0, sizeof(double), smt_stream(stream)));
    if (n == 0) return SMT_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_sum_f32, dim3(blocks), dim3(256), 0, smt_stream(stream), x, n, out_dev);
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

SMT_API int smt_median_filter(const float *in, float *out, int W, int H, int wnd_size, void *stream)
{
    if (!in || !out || in == out || H <= 0 || W <= 0 || wnd_size < 1 || wnd_size > 7) return SMT_ERR_ARG;
    const int n = H * W;
    hipLaunchKernelGGL(k_median, dim3((n + NT - 1) / NT), dim3(NT), 0, smt_stream(stream), in, out, W, H,
                       wnd_size / 2);
    SMT_LAUNCH_CHECK();
    return SMT_OK;
}

// ---------------------------------------------------------------------------------------------
// RemoveSpeckles (AD-CensusV1/PostProcessing.h:250-311): 8-connected regions of pixels whose
// neighbouring disparities differ by <= diff_insame; regions smaller than min_speckle_aera are
// set to invalid_val.  The reference grows regions by BFS in scan order, but membership is the
// transitive closure of a SYMMETRIC relation on the unmodified input (pixels it invalidates were
// already visited), so the partition is order-independent: connected-component labelling by
// min-label propagation + pointer jumping, saturating size counts, then the invalidation pass.
// `invalid_val` is an int as in the reference's signature (`const int&`; its call sites pass
// +inf, whose int conversion is undefined -- INT_MIN on x86).
namespace {

__device__ __forceinline__ bool sp_linked(float a, float b, float inv, float diff)
{
    return a != inv && b != inv && fabsf(b - a) <= diff;                  // :290-292
}

__global__ void __launch_bounds__(NT) k_cc_init(int n, int *label, int *count)
{
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p < n) { label[p] = p; count[p] = 0; }
}

__global__ void __launch_bounds__(NT) k_cc_scan(const float *__restrict__ d, int W, int H, float inv, float diff,
                                                int *label, int *changed)
{
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= W * H) return;
    const float dp = d[p];
    if (dp == inv) return;
    const int i = p / W, j = p - i * W;
    const int lp = label[p];
    int m = lp;
    for (int r = -1; r <= 1; r++)
        for (int c = -1; c <= 1; c++) {
            if (r == 0 && c == 0) continue;
            const int ii = i + r, jj = j + c;
            if (ii < 0 || ii >= H || jj < 0 || jj >= W) continue;
            const int q = ii * W + jj;
            if (sp_linked(dp, d[q], inv, diff)) m = min(m, label[q]);
        }
    if (m < lp) {
        atomicMin(&label[lp], m);          // hook this pixel's current root under the smaller label
        atomicMin(&label[p], m);
        *changed = 1;
    }
}

__global__ void __launch_bounds__(NT) k_cc_jump(int n, int *label)
{
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= n) return;
    int l = label[p];
    while (label[l] != l) l = label[l];
    label[p] = l;
}

__global__ void __launch_bounds__(NT) k_cc_count(const float *__restrict__ d, int n, float inv, const int *label,
                                                 int *count, int cap)
{
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= n || d[p] == inv) return;
    const int l = label[p];
    if (*(volatile int *)&count[l] < cap) atomicAdd(&count[l], 1);     // saturating: only "< min area?" matters
}

__global__ void __launch_bounds__(NT) k_cc_apply(float *d, int n, float inv, const int *label, const int *count,
                                                 int min_area)
{
    const int p = blockIdx.x * NT + threadIdx.x;
    if (p >= n || d[p] == inv) return;
    if ((unsigned)count[label[p]] < (unsigned)min_area) d[p] = inv;     // :304-308
}

}  // namespace

SMT_API int smt_remove_speckles(float *disp, int W, int H, int diff_insame, unsigned min_speckle_area,
                                int invalid_val, void *stream)
{
    if (!disp || W <= 0 || H <= 0) return SMT_ERR_ARG;
    hipStream_t st = smt_stream(stream);
    const int n = W * H;
    int *label = nullptr, *count = nullptr, *flag = nullptr;
    int rc = smt_malloc((void **)&label, (size_t)n * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&count, (size_t)n * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&flag, 4);
    if (rc != SMT_OK) { (void)hipFree(label); (void)hipFree(count); (void)hipFree(flag); return rc; }
    const float inv = (float)invalid_val, diff = (float)diff_insame;
    const dim3 grid((n + NT - 1) / NT);
    hipLaunchKernelGGL(k_cc_init, grid, dim3(NT), 0, st, n, label, count);
    rc = SMT_OK;
    for (int it = 0; it < n; it++) {                  // converges in far fewer rounds; n bounds it
        int h = 0;
        if (hipMemsetAsync(flag, 0, 4, st) != hipSuccess) { rc = SMT_ERR_HIP; break; }
        hipLaunchKernelGGL(k_cc_scan, grid, dim3(NT), 0, st, disp, W, H, inv, diff, label, flag);
        hipLaunchKernelGGL(k_cc_jump, grid, dim3(NT), 0, st, n, label);
        if (hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = SMT_ERR_HIP; break; }
        if (!h) break;
    }
    if (rc == SMT_OK) {
        const int cap = min_speckle_area > 0x7fffffffu ? 0x7fffffff : (int)min_speckle_area;
        hipLaunchKernelGGL(k_cc_count, grid, dim3(NT), 0, st, disp, n, inv, label, count, cap);
        hipLaunchKernelGGL(k_cc_apply, grid, dim3(NT), 0, st, disp, n, inv, label, count, cap);
        if (hipStreamSynchronize(st) != hipSuccess) rc = SMT_ERR_HIP;
    }
    (void)hipFree(label); (void)hipFree(count); (void)hipFree(flag);
    return rc;
}
