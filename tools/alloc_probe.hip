// Does the store-only rate of the cost kernel's pattern depend on WHICH device allocation it writes to?
// Allocates the two 1080p x 192 volumes repeatedly (keeping earlier ones alive so that later trials get
// different physical pages, then freeing everything and starting over) and times the XCD-contiguous
// streaming-store pattern on each.  Also prints the in-kernel shader clock (s_memtime / s_memrealtime).
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/alloc_probe.hip -o /tmp/ap && /tmp/ap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int C>
__global__ void __launch_bounds__(256) k_rows_xcd(float *vol0, float *vol1, int H, int W, int nbx,
                                                  unsigned long long *stamp)
{
    constexpr int D = 64 * C, FTJ = 64, FPW = 16;
    typedef float fvec __attribute__((ext_vector_type(C == 3 ? 3 : C), aligned(4)));
    const long nb = (long)nbx * H * 2, per = (nb + 7) / 8;
    const long b = blockIdx.x;
    const long c = (b & 7) * per + (b >> 3);
    if (c >= nb) return;
    unsigned long long t0 = 0, r0 = 0;
    if (stamp) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
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
        __builtin_nontemporal_store(x, reinterpret_cast<fvec *>(out));
        out += D;
        x[0] += 1.0f;
    }
    if (stamp && threadIdx.x == 0) {
        unsigned long long *s = stamp + 4 * (size_t)blockIdx.x;
        s[0] = t0; s[1] = r0; s[2] = __builtin_amdgcn_s_memtime(); s[3] = __builtin_amdgcn_s_memrealtime();
    }
}

int main()
{
    const int H = 1080, W = 1920, D = 192, nbx = (W + 63) / 64;
    const size_t V = (size_t)H * W * D;
    const unsigned nblk = (unsigned)(((long)nbx * H * 2 + 7) / 8 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long *stamp; hipMalloc(&stamp, (size_t)nblk * 32);
    std::vector<unsigned long long> hs((size_t)nblk * 4);
    for (int round = 0; round < 3; round++) {
        std::vector<float *> keep;
        for (int trial = 0; trial < 12; trial++) {
            float *a = nullptr, *b = nullptr;
            if (hipMalloc(&a, V * 4) != hipSuccess || hipMalloc(&b, V * 4) != hipSuccess) { printf("alloc failed\n"); break; }
            keep.push_back(a); keep.push_back(b);
            float ms;
            for (int k = 0; k < 5; k++) hipLaunchKernelGGL(k_rows_xcd<3>, dim3(nblk), dim3(256), 0, 0, a, b, H, W, nbx, (unsigned long long *)nullptr);
            hipEventRecord(e0);
            for (int k = 0; k < 40; k++) hipLaunchKernelGGL(k_rows_xcd<3>, dim3(nblk), dim3(256), 0, 0, a, b, H, W, nbx, (unsigned long long *)nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            hipLaunchKernelGGL(k_rows_xcd<3>, dim3(nblk), dim3(256), 0, 0, a, b, H, W, nbx, stamp);
            hipMemcpy(hs.data(), stamp, (size_t)nblk * 32, hipMemcpyDeviceToHost);
            std::vector<float> ratio;
            for (unsigned k = 0; k < nblk; k++) {
                const unsigned long long dt = hs[4 * k + 2] - hs[4 * k], dr = hs[4 * k + 3] - hs[4 * k + 1];
                if (hs[4 * k + 3] && dr >= 100) ratio.push_back((float)((double)dt / dr * 100.0));
            }
            float mhz = 0;
            if (!ratio.empty()) { std::nth_element(ratio.begin(), ratio.begin() + ratio.size() / 2, ratio.end()); mhz = ratio[ratio.size() / 2]; }
            // each volume by itself, written as both "views" of a pair-sized launch split in two halves:
            // (a-first-half, a-second-half) -- is slow/fast a property of one buffer or of the pair?
            float msa, msb;
            float *a2 = a + V / 2, *b2 = b + V / 2;
            hipEventRecord(e0);
            for (int k = 0; k < 40; k++) hipLaunchKernelGGL(k_rows_xcd<3>, dim3(nblk / 2), dim3(256), 0, 0, a, a2, H / 2, W, nbx, (unsigned long long *)nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msa, e0, e1);
            hipEventRecord(e0);
            for (int k = 0; k < 40; k++) hipLaunchKernelGGL(k_rows_xcd<3>, dim3(nblk / 2), dim3(256), 0, 0, b, b2, H / 2, W, nbx, (unsigned long long *)nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msb, e0, e1);
            printf("round %d trial %2d  a=%p b=%p  store-only %.4f ms  %.0f GB/s  sclk %.0f MHz  a-alone %.4f b-alone %.4f (x2 for the pair)\n", round, trial, (void *)a, (void *)b,
                   ms / 40, 2.0 * V * 4 / (ms / 40 * 1e-3) / 1e9, mhz, msa / 40, msb / 40);
            fflush(stdout);
        }
        for (float *p : keep) hipFree(p);
    }
    return 0;
}
