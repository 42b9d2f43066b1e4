// Counterpart of AD-CensusV1/main.cpp:13-121 on a synthetic pair: same call order
// (AD_Census -> CrossArmAggregation L/R -> ScanlineOptimizer on the left volume -> WTA ->
// LeftRightConsistency), host buffers in and out, everything computed by libsmt_hip.so.
// Prints FNV-1a hashes of every product so tests can compare them with the oracle's.
//   usage: adcensus_main H W D seed
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "smt_host.hpp"

static double g_hash_ms = 0;   // time spent hashing products, excluded from the pipeline figure
static uint64_t fnv(const void *p, size_t n)
{
    auto t = std::chrono::steady_clock::now();
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t k = 0; k < n; k++) { h ^= b[k]; h *= 1099511628211ull; }
    g_hash_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
    return h;
}

// integer-only synthetic pair, SURVEY.md 8d (same generator as synth.py)
static int tri(int x, int p) { int m = x % (2 * p); int v = m < p ? m : 2 * p - m; return v - p / 2; }
static void synth(int H, int W, int D, uint32_t seed, std::vector<unsigned char> &L, std::vector<unsigned char> &R)
{
    uint32_t s = seed;
    L.resize((size_t)H * W); R.resize((size_t)H * W);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            s = s * 1664525u + 1013904223u;
            int b = (int)(s >> 24);
            int v = 128 + tri(j, 203) * 70 / 101 + tri(i, 139) * 40 / 69 + 25 * (((j / 40) + (i / 30)) & 1) + (b % 6);
            R[(size_t)i * W + j] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    for (int i = 0; i < H; i++) {
        int g = D / 8 + ((i / 8) % 7) * (D / 16);
        for (int j = 0; j < W; j++) {
            s = s * 1664525u + 1013904223u;
            L[(size_t)i * W + j] = j >= g ? R[(size_t)i * W + j - g] : (unsigned char)(s >> 24);
        }
    }
}

int main(int argc, char **argv)
{
    const int row = argc > 1 ? atoi(argv[1]) : 72, col = argc > 2 ? atoi(argv[2]) : 160;
    const int dispRange = argc > 3 ? atoi(argv[3]) : 64;
    const uint32_t seed = argc > 4 ? (uint32_t)atoi(argv[4]) : 3;
    const float sigmaS = 30, sigmaC = 10;                 // main.cpp:25-26
    const int tao = 30, p1 = 10, p2 = 150, gate = 2;      // main.cpp:27-30
    try {
        std::vector<unsigned char> leftGray, rightGray;
        synth(row, col, dispRange, seed, leftGray, rightGray);
        const size_t n = (size_t)row * col, V = n * dispRange;
        std::vector<float> leftptr(n), rightptr(n), leftDisp(n), rightDisp(n);
        for (size_t k = 0; k < n; k++) { leftptr[k] = leftGray[k]; rightptr[k] = rightGray[k]; }   // main.cpp:46-55
        std::vector<float> aggL(V), aggR(V);
        { void *w = nullptr; if (smt_malloc(&w, 256) == SMT_OK) smt_free(w); }   // HIP context creation stays outside the timer
        auto t0 = std::chrono::steady_clock::now();

        smt::AD_Census ADcensus;
        ADcensus.Initialize(leftptr.data(), rightptr.data(), dispRange, row, col, sigmaC, sigmaS);
        ADcensus.ComputeADcensus();
        ADcensus.ComputeADcensusRight();
        ADcensus.WTA(leftDisp.data(), rightDisp.data());
        float *costVolumeLeftPtr = ADcensus.GetPtrLeft();
        float *costVolumeRightPtr = ADcensus.GetPtrRight();
        printf("cost_left %016llx\ncost_right %016llx\n", (unsigned long long)fnv(costVolumeLeftPtr, V * 4),
               (unsigned long long)fnv(costVolumeRightPtr, V * 4));
        printf("wta_left %016llx\nwta_right %016llx\n", (unsigned long long)fnv(leftDisp.data(), n * 4),
               (unsigned long long)fnv(rightDisp.data(), n * 4));

        smt::CrossArmAggregation CrossArm;
        CrossArm.Initialize(row, col, leftptr.data(), rightptr.data(), tao, dispRange);
        CrossArm.ComputeArmLengths(leftGray.data(), 1);
        CrossArm.AggregationVertical(costVolumeLeftPtr, aggL.data());
        CrossArm.WTA(aggL.data(), leftDisp.data());
        CrossArm.Initialize(row, col, leftptr.data(), rightptr.data(), tao, dispRange);
        CrossArm.ComputeArmLengths(rightGray.data(), 1);
        CrossArm.AggregationVertical(costVolumeRightPtr, aggR.data());
        CrossArm.WTA(aggR.data(), rightDisp.data());
        printf("agg_left %016llx\nagg_right %016llx\n", (unsigned long long)fnv(aggL.data(), V * 4),
               (unsigned long long)fnv(aggR.data(), V * 4));

        smt::ScanlineOptimizer ScanlineOpt;               // main.cpp:86-89 (enabled)
        ScanlineOpt.Initialize(row, col, dispRange, aggL.data(), p1, p2);
        ScanlineOpt.ScanLine(aggL.data(), leftptr.data());
        ScanlineOpt.WTA(leftDisp.data());
        printf("so_wta_left %016llx\n", (unsigned long long)fnv(leftDisp.data(), n * 4));

        std::vector<std::pair<int, int>> occlusions, mismatches;
        smt::LeftRightConsistency(col, row, gate, leftDisp.data(), rightDisp.data(), occlusions, mismatches);   // main.cpp:92
        printf("lr_left %016llx\nn_occlusion %zu\nn_mismatch %zu\n", (unsigned long long)fnv(leftDisp.data(), n * 4),
               occlusions.size(), mismatches.size());
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - g_hash_ms;
        fprintf(stderr, "host-buffer pipeline (PCIe-inclusive) %.2f ms for %dx%d D=%d\n", ms, col, row, dispRange);
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
