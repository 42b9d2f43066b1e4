// Micro-benchmark: what HBM write rate does the cost kernel's store pattern reach with NO compute?
// Same grid as k_cost_fast2 at 1920x1080, D=192 (30 x 1080 x 2 workgroups of 4 waves; each wave writes 16
// consecutive pixels x 768 B with one dwordx3 per lane), vs a plain float4 grid-stride fill of the same bytes.
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/store_ceiling.hip -o /tmp/sc && /tmp/sc
#include <hip/hip_runtime.h>
#include <cstdio>

struct f3 { float v[3]; };

__global__ void __launch_bounds__(256) k_pattern(float *vol0, float *vol1, int H, int W)
{
    constexpr int D = 192, FTJ = 64, FPW = 16;
    float *vol = blockIdx.z ? vol1 : vol0;
    const int i = blockIdx.y, j0 = blockIdx.x * FTJ;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int p0 = wid * FPW;
    float *out = vol + ((size_t)i * W + j0 + p0) * D + lane * 3;
    const int npx = min(FPW, W - (j0 + p0));
    f3 x = {{(float)lane, 1.0f, 2.0f}};
    for (int q = 0; q < npx; q++) {
        *reinterpret_cast<f3 *>(out) = x;
        out += D;
        x.v[0] += 1.0f;
    }
}

__global__ void __launch_bounds__(256) k_fill4(float4 *p, size_t n)
{
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256)
        p[k] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main()
{
    const int H = 1080, W = 1920, D = 192;
    const size_t V = (size_t)H * W * D;
    float *a, *b;
    hipMalloc(&a, V * 4); hipMalloc(&b, V * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int k = 0; k < 20; k++) hipLaunchKernelGGL(k_pattern, dim3(30, H, 2), dim3(256), 0, 0, a, b, H, W);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("pattern   : %.4f ms per pair-equivalent, %.1f GB/s\n", ms / 20, 2.0 * V * 4 / (ms / 20 * 1e-3) / 1e9);
        hipEventRecord(e0);
        for (int k = 0; k < 20; k++) {
            hipLaunchKernelGGL(k_fill4, dim3(2048), dim3(256), 0, 0, (float4 *)a, V / 4);
            hipLaunchKernelGGL(k_fill4, dim3(2048), dim3(256), 0, 0, (float4 *)b, V / 4);
        }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("float4 fill: %.4f ms per pair-equivalent, %.1f GB/s\n", ms / 20, 2.0 * V * 4 / (ms / 20 * 1e-3) / 1e9);
    }
    return 0;
}
