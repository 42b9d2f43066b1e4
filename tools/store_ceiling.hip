// Micro-benchmark: what HBM write rate does the cost kernel's store pattern reach with NO compute?
// Same grid as k_cost_fast2 (ceil(W/64) x H x 2 workgroups of 4 waves; each wave writes 16 consecutive
// pixels x D*4 B with one dword{x2,x3,x4} per lane = "rows", in dispatch order and in the XCD-contiguous
// order the kernel uses, with ordinary and with non-temporal stores), against two alternatives over the same
// bytes: the wave's 16*D*4 contiguous bytes written as flat 1 KB dwordx4 stores ("flat"), and a plain
// float4 grid-stride fill.
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/store_ceiling.hip -o /tmp/sc && /tmp/sc
#include <hip/hip_runtime.h>
#include <cstdio>

template <int C> struct fv { float v[C]; };
template <> struct __attribute__((aligned(8))) fv<2> { float v[2]; };
template <> struct __attribute__((aligned(16))) fv<4> { float v[4]; };

template <int C>
__global__ void __launch_bounds__(256) k_rows(float *vol0, float *vol1, int H, int W)
{
    constexpr int D = 64 * C, FTJ = 64, FPW = 16;
    float *vol = blockIdx.z ? vol1 : vol0;
    const int i = blockIdx.y, j0 = blockIdx.x * FTJ;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int p0 = wid * FPW;
    float *out = vol + ((size_t)i * W + j0 + p0) * D + lane * C;
    const int npx = min(FPW, W - (j0 + p0));
    fv<C> x;
    for (int k = 0; k < C; k++) x.v[k] = (float)(lane + k);
    for (int q = 0; q < npx; q++) {
        *reinterpret_cast<fv<C> *>(out) = x;
        out += D;
        x.v[0] += 1.0f;
    }
}

// the same stores with the workgroup -> chunk order of k_cost_fast2: workgroup b runs on XCD b % 8 and
// takes chunk (b % 8) * ceil(n / 8) + b / 8, i.e. every XCD streams one contiguous eighth of the volumes
template <int C, bool NT>
__global__ void __launch_bounds__(256) k_rows_xcd(float *vol0, float *vol1, int H, int W, int nbx)
{
    constexpr int D = 64 * C, FTJ = 64, FPW = 16;
    typedef float fvec __attribute__((ext_vector_type(C == 3 ? 3 : C), aligned(4)));
    const long nb = (long)nbx * H * 2, per = (nb + 7) / 8;
    const long b = blockIdx.x;
    const long c = (b & 7) * per + (b >> 3);
    if (c >= nb) return;
    const int z = (int)(c / ((long)nbx * H));
    const long r = c - (long)z * nbx * H;
    const int i = (int)(r / nbx), bx = (int)(r - (long)i * nbx);
    float *vol = z ? vol1 : vol0;
    const int j0 = bx * FTJ;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int p0 = wid * FPW;
    float *out = vol + ((size_t)i * W + j0 + p0) * D + lane * C;
    const int npx = min(FPW, W - (j0 + p0));
    fvec x;
    for (int k = 0; k < C; k++) x[k] = (float)(lane + k);
    for (int q = 0; q < npx; q++) {
        if (NT) __builtin_nontemporal_store(x, reinterpret_cast<fvec *>(out));
        else *reinterpret_cast<fvec *>(out) = x;
        out += D;
        x[0] += 1.0f;
    }
}

template <int C>
__global__ void __launch_bounds__(256) k_flat(float *vol0, float *vol1, int H, int W)
{
    constexpr int D = 64 * C, FTJ = 64, FPW = 16;
    float *vol = blockIdx.z ? vol1 : vol0;
    const int i = blockIdx.y, j0 = blockIdx.x * FTJ;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int p0 = wid * FPW;
    const int npx = min(FPW, W - (j0 + p0));
    float4 *out = reinterpret_cast<float4 *>(vol + ((size_t)i * W + j0 + p0) * D) + lane;
    float4 x = make_float4((float)lane, 1.f, 2.f, 3.f);
    const int n4 = npx * D / 4;                        // float4 per wave region
    for (int q = lane; q < n4; q += 64) {
        *out = x;
        out += 64;
        x.x += 1.0f;
    }
}

__global__ void __launch_bounds__(256) k_fill4(float4 *p, size_t n)
{
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256)
        p[k] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <int C>
static void run(int H, int W)
{
    const int D = 64 * C;
    const size_t V = (size_t)H * W * D;
    float *a, *b;
    hipMalloc(&a, V * 4); hipMalloc(&b, V * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    const dim3 grid((W + 63) / 64, H, 2);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int k = 0; k < 20; k++) hipLaunchKernelGGL(k_rows<C>, grid, dim3(256), 0, 0, a, b, H, W);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("%dx%d D=%d rows       : %.4f ms per pair-equivalent, %.1f GB/s\n", W, H, D, ms / 20, 2.0 * V * 4 / (ms / 20 * 1e-3) / 1e9);
        {
            const int nbx = (W + 63) / 64;
            const dim3 g1((unsigned)(((long)nbx * H * 2 + 7) / 8 * 8));
            hipEventRecord(e0);
            for (int k = 0; k < 20; k++) hipLaunchKernelGGL((k_rows_xcd<C, false>), g1, dim3(256), 0, 0, a, b, H, W, nbx);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("%dx%d D=%d rows, XCD-contiguous     : %.4f ms per pair-equivalent, %.1f GB/s\n", W, H, D, ms / 20, 2.0 * V * 4 / (ms / 20 * 1e-3) / 1e9);
            hipEventRecord(e0);
            for (int k = 0; k < 20; k++) hipLaunchKernelGGL((k_rows_xcd<C, true>), g1, dim3(256), 0, 0, a, b, H, W, nbx);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("%dx%d D=%d rows, XCD-contiguous, nt : %.4f ms per pair-equivalent, %.1f GB/s\n", W, H, D, ms / 20, 2.0 * V * 4 / (ms / 20 * 1e-3) / 1e9);
        }
        hipEventRecord(e0);
        for (int k = 0; k < 20; k++) hipLaunchKernelGGL(k_flat<C>, grid, dim3(256), 0, 0, a, b, H, W);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("%dx%d D=%d flat 1 KB  : %.4f ms per pair-equivalent, %.1f GB/s\n", W, H, D, ms / 20, 2.0 * V * 4 / (ms / 20 * 1e-3) / 1e9);
        hipEventRecord(e0);
        for (int k = 0; k < 20; k++) {
            hipLaunchKernelGGL(k_fill4, dim3(2048), dim3(256), 0, 0, (float4 *)a, V / 4);
            hipLaunchKernelGGL(k_fill4, dim3(2048), dim3(256), 0, 0, (float4 *)b, V / 4);
        }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("%dx%d D=%d float4 fill: %.4f ms per pair-equivalent, %.1f GB/s\n", W, H, D, ms / 20, 2.0 * V * 4 / (ms / 20 * 1e-3) / 1e9);
    }
    hipFree(a); hipFree(b);
}

int main(int argc, char **argv)
{
    run<3>(1080, 1920);
    run<2>(720, 1280);
    run<4>(375, 1242);
    if (argc > 1) {                                    // size vs row-stride: which one lowers the 1080p D=192 rate?
        run<2>(1080, 1920);
        run<4>(1080, 1920);
        run<3>(720, 1280);
        run<3>(375, 1242);
    }
    return 0;
}
