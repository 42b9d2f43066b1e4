// Can the rectangle aggregation's flagged accumulate  acc[q] = fma(x, member(q) ? 1 : 0, acc[q])  run on the matrix
// pipe?  v_mfma_f32_4x4x1_16b_f32 computes, per 4-lane block b, D_b[i][j] = A_b[i] * B_b[j] + C_b[i][j]: with
// A = the membership flags of 4 pixels (lane l holds the flag of pixel l % 4) and B = the tap's row (lane l holds
// hypothesis l), accumulator register i of lane l becomes acc_i[l] + flag_i * x[l] -- four pixels x 64 hypotheses
// per instruction, in the accumulator layout the kernel already has.
//  1. exactness: T taps of random x (normal, tiny, denormal, negative, zero) under random flags through the MFMA
//     chain and through v_fma_f32 / v_add_f32 in the same order: the bits must be equal;
//  2. rate: cycles per MFMA per SIMD, back to back on independent accumulators, 1..4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_flag_probe tools/mfma_flag_probe.hip && /tmp/mfma_flag_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) k_exact(const float *__restrict__ x, const unsigned *__restrict__ mask, int T,
                                              float *__restrict__ out_mfma, float *__restrict__ out_fma, float *__restrict__ out_add)
{
    const int lane = threadIdx.x;
    f4 acc = f4{0.0f, 0.0f, 0.0f, 0.0f};
    float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    for (int t = 0; t < T; t++) {
        const float xv = x[(size_t)t * 64 + lane];
        const unsigned m = mask[t];                                   // 4 membership bits
        const float A = ((m >> (lane & 3)) & 1u) ? 1.0f : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A, xv, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float f = ((m >> i) & 1u) ? 1.0f : 0.0f;
            a1[i] = __builtin_fmaf(xv, f, a1[i]);                      // what the kernel does today
            if ((m >> i) & 1u) a2[i] = a2[i] + xv;                     // the reference's own add
        }
    }
    for (int i = 0; i < 4; i++) {
        out_mfma[i * 64 + lane] = acc[i];
        out_fma[i * 64 + lane] = a1[i];
        out_add[i * 64 + lane] = a2[i];
    }
}

template <int NACC>
__global__ void __launch_bounds__(1024) k_rate(uint64_t *out, float seed, int iters)
{
    f4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f4{seed, seed + 1, seed + 2, seed + 3};
    const float A = (threadIdx.x & 1) ? 1.0f : 0.0f, x = seed + threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, x, acc[i], 0, 0, 0);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int NACC>
static void rate(uint64_t *d)
{
    printf("v_mfma_f32_4x4x1_16b_f32, %d independent accumulators:", NACC);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = wps >= 4 ? 1024 : 256 * wps, blocks = 256;
        const int iters = 64, big = 4096;
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_rate<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f, iters);
        (void)hipDeviceSynchronize();
        uint64_t h = 0;
        (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f, big);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %dw: %5.2f cyc %5.2f ns", wps, (double)h / (iters * 16.0 * NACC) / wps, ms * 1e6 / ((double)big * 16 * NACC * wps));
    }
    printf("  (per MFMA per SIMD)\n");
}

int main()
{
    const int T = 4096;
    std::vector<float> x((size_t)T * 64);
    std::vector<unsigned> m(T);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (size_t k = 0; k < x.size(); k++) {
        const unsigned r = rnd();
        float v;
        switch ((r >> 28) & 7) {
        case 0: v = (float)(r & 0xffff) / 65536.0f * 2.0f; break;                       // costs in [0, 2)
        case 1: v = (float)(r & 0xffffff) * 1e-3f; break;
        case 2: { uint32_t b = (r & 0x007fffffu) | 0x00000000u; memcpy(&v, &b, 4); break; }   // denormal
        case 3: { uint32_t b = (r & 0x807fffffu) | 0x00800000u; memcpy(&v, &b, 4); break; }   // smallest normals, both signs
        case 4: v = -(float)(r & 0xffff) / 1024.0f; break;
        case 5: v = 0.0f; break;
        case 6: { uint32_t b = (r & 0x7fffffu) | ((100u + (r >> 20) % 60u) << 23) | (r & 0x80000000u); memcpy(&v, &b, 4); break; }
        default: v = (float)(int)(r & 0xff); break;
        }
        x[k] = v;
    }
    for (int t = 0; t < T; t++) m[t] = (rnd() >> 13) & 15u;
    float *dx, *o1, *o2, *o3;
    unsigned *dm;
    uint64_t *dr;
    (void)hipMalloc(&dx, x.size() * 4); (void)hipMalloc(&dm, T * 4); (void)hipMalloc(&o1, 1024); (void)hipMalloc(&o2, 1024); (void)hipMalloc(&o3, 1024);
    (void)hipMalloc(&dr, 16);
    (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dm, m.data(), T * 4, hipMemcpyHostToDevice);
    for (int Tt : {1, 7, 64, 1000, 4096}) {
        hipLaunchKernelGGL(k_exact, dim3(1), dim3(64), 0, 0, dx, dm, Tt, o1, o2, o3);
        uint32_t a[256], b[256], c[256];
        (void)hipMemcpy(a, o1, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(b, o2, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(c, o3, 1024, hipMemcpyDeviceToHost);
        int d_fma = 0, d_add = 0, fma_add = 0;
        for (int k = 0; k < 256; k++) { d_fma += a[k] != b[k]; d_add += a[k] != c[k]; fma_add += b[k] != c[k]; }
        printf("exactness, %4d taps: MFMA vs v_fma chain %d / 256 differ, MFMA vs plain adds %d / 256, v_fma chain vs plain adds %d / 256\n", Tt, d_fma, d_add, fma_add);
    }
    rate<1>(dr); rate<3>(dr); rate<6>(dr); rate<12>(dr);
    return 0;
}
