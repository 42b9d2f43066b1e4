// Issue cost (cycles per wave64 instruction on one SIMD) of the VALU instructions the ASW tap loop is made of.
// 1, 2 and 4 waves per SIMD (one workgroup per CU), 8 independent chains, s_memtime around 64 x 8 x 16 instructions.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(1024) k_rate(uint64_t *out, double seed, int iters)
{
    double a[8], b = seed + threadIdx.x, c = seed * 0.5;
    float f[8];
    unsigned u[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i; f[i] = (float)seed + i; u[i] = (unsigned)(seed) + i + threadIdx.x; }
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#define X(i)                                                                                          \
    if (OP == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));             \
    if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));                         \
    if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));                         \
    if (OP == 3) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));                         \
    if (OP == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));                      \
    if (OP == 5) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));                      \
    if (OP == 6) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));            \
    if (OP == 7) asm volatile("v_min_f32 %0, |%0|, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));          \
    if (OP == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                      \
    if (OP == 9) asm volatile("v_sad_u16 %0, %0, %1, 0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));         \
    if (OP == 10) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(u[i]));                     \
    if (OP == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                     \
    if (OP == 12) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7])); \
    if (OP == 13) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));                        \
    if (OP == 14) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));       \
    if (OP == 15) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(u[i]));
            REP8(X)
#undef X
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + f[i] + u[i];
    if (s == 12345.678) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int OP>
static void run(const char *name, uint64_t *d)
{
    const int iters = 64;
    printf("%-16s", name);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int big = 4096;                                // iterations of the event-timed launch
    for (int wps = 1; wps <= 16; wps *= 2) {             // waves per SIMD: 256 * wps / 1024 workgroups of 16 waves per CU
        const int threads = wps >= 4 ? 1024 : 256 * wps, blocks = wps >= 4 ? 256 * (wps / 4) : 256;
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1.0, iters);
        hipDeviceSynchronize();
        uint64_t h = 0;
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1.0, big);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // s_memtime counts core clocks; the event time gives ns per instruction per SIMD
        printf("  %2d w/SIMD: %.2f cyc (%.2f ns)", wps, (double)h / (iters * 16 * 8) / wps, ms * 1e6 / ((double)big * 16 * 8 * wps));
    }
    printf("\n");
}

int main()
{
    uint64_t *d;
    hipMalloc(&d, 16);
    run<0>("v_fma_f64", d); run<1>("v_mul_f64", d); run<2>("v_add_f64", d); run<3>("v_min_f64", d);
    run<13>("v_max_f64", d); run<15>("v_ldexp_f64", d);
    run<4>("v_cvt_f64_f32", d); run<5>("v_cvt_f64_u32", d); run<10>("v_cvt_f64_i32", d);
    run<6>("v_sub_f32", d); run<7>("v_min_f32|.|", d); run<12>("v_med3_f32", d); run<14>("v_fma_f32", d);
    run<8>("v_pk_add_f32", d); run<11>("v_pk_mul_f32", d); run<9>("v_sad_u16", d);
    run<0>("v_fma_f64", d);
    return 0;
}
