// Two issue-rate questions, answered on the device (gfx950), one number per cell:
//  A. fp32 accumulate forms the rectangle aggregation could use (csrc/crossarm.hip): is v_pk_fma_f32 worth two
//     v_fma_f32 on a SIMD-32 part, with VGPR and with SGPR multipliers?  cycles per wave64 instruction per SIMD at
//     1..8 waves per SIMD (s_memtime) and ns per instruction per SIMD (events).
//  C. the same question for the f32 matrix instruction the aggregation variants use, beside v_pk_fma_f32.
//  B. does the f64 matrix pipe run BESIDE the f64 vector pipe (ASW denominator on v_mfma_f64_16x16x4_f64 next to the
//     v_fma_f64 numerator)?  MFMA-only, VALU-only, both interleaved in one wave's stream, and
//     both in different waves of a SIMD: if the pipes overlap, "both" costs max(), else the sum.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_probe tools/issue_probe.hip && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <type_traits>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

template <int OP>
__global__ void __launch_bounds__(1024) k_f32(uint64_t *out, float seed, int iters)
{
    float f[8], x = seed + threadIdx.x;
    f2 p[8], px = f2{x, x + 1.0f};
    for (int i = 0; i < 8; i++) { f[i] = seed + i; p[i] = f2{seed + i, seed - i}; }
    float s0 = 1.0f, s1 = 0.0f;
    asm volatile("s_mov_b32 %0, 1.0" : "=s"(s0));
    asm volatile("s_mov_b32 %0, 0" : "=s"(s1));
    uint64_t sp;                                          // {1.0f, 0.0f} as an SGPR pair
    asm volatile("s_mov_b64 %0, 0x3f800000" : "=s"(sp));
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#define X(i)                                                                                                  \
    if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(x), "v"(f[(i + 1) & 7]));        \
    if (OP == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(x), "s"(s0));                     \
    if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(x));                                  \
    if (OP == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(px), "v"(p[(i + 1) & 7]));     \
    if (OP == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(px), "s"(sp)); \
    if (OP == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(px));                              \
    if (OP == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[i]) : "s"(s0), "v"(x));                        \
    if (OP == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(x));
            REP8(X)
#undef X
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = s1;
    for (int i = 0; i < 8; i++) s += f[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

// MODE 0: MFMA only; 1: VALU only; 2: both in every wave's stream (NV v_fma_f64 after each MFMA); 3: even waves MFMA
// only, odd waves VALU only (wave id = threadIdx.x / 64; waves w and w + 4 share a SIMD, so pairs (w, w + 4) are
// made of one of each by using bit 2 of the wave id)
template <int MODE, int NV>
__global__ void __launch_bounds__(512) k_f64(uint64_t *out, double seed, int iters)
{
    const int wv = threadIdx.x >> 6;
    double a = seed + (threadIdx.x & 63), b = seed * 0.5 + 1.0;
    d4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = d4{seed, seed + 1, seed + 2, seed + 3};
    double v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + i;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && !(wv & 4));
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wv & 4));
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (do_m) acc[r & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r & 3], 0, 0, 0);
            if (do_v) {
#pragma unroll
                for (int k = 0; k < NV; k++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[k & 7]) : "v"(a), "v"(b));
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; i++) s += v[i];
    if (s == 12345.678) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

// Part C: the f32 matrix instruction the aggregation variants 8 / 9 / 11 use (v_mfma_f32_4x4x1_16b_f32, 256 FMAs) beside
// v_pk_fma_f32 (128 FMAs).  Same MODE meaning as k_f64; one step = 1 MFMA and / or NV packed FMAs.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
template <int MODE, int NV>
__global__ void __launch_bounds__(512) k_f32m(uint64_t *out, float seed, int iters)
{
    const int wv = threadIdx.x >> 6;
    float a = seed + (threadIdx.x & 63), b = seed * 0.5f + 1.0f;
    f4v acc[4];
    for (int i = 0; i < 4; i++) acc[i] = f4v{seed, seed + 1, seed + 2, seed + 3};
    f2v v[8], x = f2v{a, b}, y = f2v{b, a};
    for (int i = 0; i < 8; i++) v[i] = f2v{seed + i, seed - i};
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && !(wv & 4));
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wv & 4));
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    // one straight-line loop per kind of wave: these instructions take 4-8 cycles, a branch per instruction would be
    // what is measured
    auto loop = [&](auto m_tag, auto v_tag) {
        constexpr bool M = decltype(m_tag)::value, V = decltype(v_tag)::value;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (M) acc[r & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[r & 3], 0, 0, 0);
                if (V) {
#pragma unroll
                    for (int k = 0; k < NV; k++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 7]) : "v"(x), "v"(y));
                }
            }
        }
    };
    if (do_m && do_v) loop(std::true_type{}, std::true_type{});
    else if (do_m) loop(std::true_type{}, std::false_type{});
    else loop(std::false_type{}, std::true_type{});
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; i++) s += v[i].x + v[i].y;
    if (s == 12345.678f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int OP>
static void run32(const char *name, uint64_t *d)
{
    const int iters = 64, big = 4096;
    printf("%-34s", name);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int threads = wps >= 4 ? 1024 : 256 * wps, blocks = wps >= 4 ? 256 * (wps / 4) : 256;
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_f32<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f, iters);
        hipDeviceSynchronize();
        uint64_t h = 0;
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_f32<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f, big);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("  %dw: %5.2f cyc %5.2f ns", wps, (double)h / (iters * 16 * 8) / wps, ms * 1e6 / ((double)big * 16 * 8 * wps));
    }
    printf("\n");
}

template <int MODE, int NV>
static double run64(const char *name, uint64_t *d, int waves_per_simd)
{
    // 512-thread workgroups = 2 waves per SIMD each; one or two workgroups per CU
    const int iters = 2048, blocks = 256 * (waves_per_simd / 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k_f64<MODE, NV>), dim3(blocks), dim3(512), 0, 0, d, 1.0, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_f64<MODE, NV>), dim3(blocks), dim3(512), 0, 0, d, 1.0, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    // per loop step of one wave: one MFMA and / or NV v_fma_f64
    printf("%-58s %d w/SIMD: %8.1f ns per step per wave  (%6.1f cycles by s_memtime)\n", name, waves_per_simd,
           ms * 1e6 / ((double)iters * 8), (double)h / ((double)iters * 8));
    return ms;
}

template <int MODE, int NV>
static double run32m(const char *name, uint64_t *d, int waves_per_simd)
{
    const int iters = 4096, blocks = 256 * (waves_per_simd / 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k_f32m<MODE, NV>), dim3(blocks), dim3(512), 0, 0, d, 1.0f, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_f32m<MODE, NV>), dim3(blocks), dim3(512), 0, 0, d, 1.0f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-58s %d w/SIMD: %8.1f ns per step per wave  (%6.1f cycles by s_memtime)\n", name, waves_per_simd,
           ms * 1e6 / ((double)iters * 8), (double)h / ((double)iters * 8));
    return ms;
}

int main()
{
    uint64_t *d;
    hipMalloc(&d, 16);
    printf("A. fp32 accumulate forms: cycles per wave64 instruction per SIMD (s_memtime / waves) and ns (events)\n");
    run32<0>("v_fma_f32 v,v,v", d);
    run32<1>("v_fma_f32 v,v,s", d);
    run32<6>("v_fmac_f32 v,s,v", d);
    run32<2>("v_add_f32", d);
    run32<3>("v_pk_fma_f32 v,v,v", d);
    run32<4>("v_pk_fma_f32 v,v,s[2] op_sel_hi:[0,1,1]", d);
    run32<5>("v_pk_add_f32", d);
    run32<7>("v_mov_b32", d);
    printf("B. f64 matrix pipe beside the f64 vector pipe (one step = 1 v_mfma_f64_16x16x4_f64 and / or NV v_fma_f64)\n");
    for (int w = 2; w <= 4; w += 2) {
        if (w == 2) {
            run64<0, 0>("MFMA only", d, w); run64<1, 8>("VALU only, NV = 8", d, w); run64<1, 16>("VALU only, NV = 16", d, w);
            run64<2, 8>("both in one stream, NV = 8", d, w); run64<2, 16>("both in one stream, NV = 16", d, w);
            run64<3, 8>("MFMA waves beside VALU waves on each SIMD, NV = 8", d, w); run64<3, 16>("MFMA waves beside VALU waves on each SIMD, NV = 16", d, w);
        } else {
            run64<0, 0>("MFMA only", d, w); run64<1, 16>("VALU only, NV = 16", d, w);
            run64<2, 16>("both in one stream, NV = 16", d, w); run64<3, 16>("MFMA waves beside VALU waves on each SIMD, NV = 16", d, w);
        }
    }
    printf("C. f32 matrix pipe beside the f32 vector pipe (one step = 1 v_mfma_f32_4x4x1_16b_f32 and / or NV v_pk_fma_f32)\n");
    for (int w = 2; w <= 4; w += 2) {
        run32m<0, 0>("MFMA only", d, w); run32m<1, 2>("VALU only, NV = 2", d, w); run32m<1, 4>("VALU only, NV = 4", d, w);
        run32m<2, 2>("both in one stream, NV = 2", d, w); run32m<2, 4>("both in one stream, NV = 4", d, w);
        run32m<3, 2>("MFMA waves beside VALU waves on each SIMD, NV = 2", d, w);
        run32m<3, 4>("MFMA waves beside VALU waves on each SIMD, NV = 4", d, w);
    }
    return 0;
}
