// Counterpart of AD-CensusV1/main.cpp:13-121 on a synthetic pair: same call order
// (AD_Census -> CrossArmAggregation L/R -> ScanlineOptimizer on the left volume -> WTA ->
// LeftRightConsistency), host buffers in and out, everything computed by libsmt_hip.so.
// Prints FNV-1a hashes of every product so tests can compare them with the oracle's.
//   usage: adcensus_main H W D seed
//          adcensus_main --images left.png right.png D [disparity_out.png]     (imread -> cvtColor -> pipeline
//                                                     -> imwrite, the file path of main.cpp:16-20, :115-117)
//          adcensus_main --batch pairs H W D             (config 5 on every visible GPU, one handle per device)
//          adcensus_main --batch-rccl pairs H W D        (the same with the maps exchanged by ncclAllGather and a
//                                                         checksum ncclAllReduce over RCCL / xGMI)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define SMT_HOST_WITH_RCCL
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

static int batch_main(int pairs, int row, int col, int dispRange, bool rccl = false)
{
    const size_t n = (size_t)row * col;
    std::vector<float> L((size_t)pairs * n), R((size_t)pairs * n), dl((size_t)pairs * n), dr((size_t)pairs * n);
    std::vector<unsigned char> l8, r8;
    for (int b = 0; b < pairs; b++) {
        synth(row, col, dispRange, 1000u + (uint32_t)b, l8, r8);
        for (size_t k = 0; k < n; k++) { L[b * n + k] = l8[k]; R[b * n + k] = r8[k]; }
    }
    { void *w = nullptr; if (smt_malloc(&w, 256) == SMT_OK) smt_free(w); }
    auto t0 = std::chrono::steady_clock::now();
    int G = 0;
    double checksum = 0.0;
    if (rccl) checksum = smt::AD_Census_batch_rccl(L.data(), R.data(), pairs, dispRange, row, col, 10.0f, 30.0f, dl.data(), dr.data(), &G);
    else G = smt::AD_Census_batch_all_devices(L.data(), R.data(), pairs, dispRange, row, col, 10.0f, 30.0f, dl.data(), dr.data());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rccl) {
        double host_sum = 0.0;
        for (size_t k = 0; k < (size_t)pairs * n; k++) host_sum += dl[k];
        printf("checksum %.1f host_sum %.1f\n", checksum, host_sum);
    }
    for (int b = 0; b < pairs; b++)
        printf("pair %d wta_left %016llx wta_right %016llx\n", b, (unsigned long long)fnv(&dl[b * n], n * 4),
               (unsigned long long)fnv(&dr[b * n], n * 4));
    fprintf(stderr, "%d pairs of %dx%d D=%d on %d device(s): %.2f ms host-to-host (PCIe-inclusive)\n", pairs, col, row, dispRange, G, ms);
    return 0;
}

int main(int argc, char **argv)
{
    int row = argc > 1 ? atoi(argv[1]) : 72, col = argc > 2 ? atoi(argv[2]) : 160;
    int dispRange = argc > 3 ? atoi(argv[3]) : 64;
    const uint32_t seed = argc > 4 ? (uint32_t)atoi(argv[4]) : 3;
    const float sigmaS = 30, sigmaC = 10;                 // main.cpp:25-26
    const int tao = 30, p1 = 10, p2 = 150, gate = 2;      // main.cpp:27-30
    const bool from_files = argc > 4 && !strcmp(argv[1], "--images");
    try {
        if (argc > 5 && !strcmp(argv[1], "--batch")) return batch_main(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]));
        if (argc > 5 && !strcmp(argv[1], "--batch-rccl")) return batch_main(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), true);
        std::vector<unsigned char> leftGray, rightGray;
        if (from_files) {
            // main.cpp:16-20: imread (3-channel BGR) + cvtColor(CV_BGR2GRAY)
            const smt::Image lg = smt::cvtColorBGR2GRAY(smt::imread(argv[2])), rg = smt::cvtColorBGR2GRAY(smt::imread(argv[3]));
            if (lg.rows != rg.rows || lg.cols != rg.cols) throw std::runtime_error("left / right sizes differ");
            row = lg.rows; col = lg.cols; dispRange = atoi(argv[4]);
            leftGray = lg.data; rightGray = rg.data;
            printf("gray_left %016llx\ngray_right %016llx\n", (unsigned long long)fnv(leftGray.data(), leftGray.size()),
                   (unsigned long long)fnv(rightGray.data(), rightGray.size()));
        } else
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
        if (from_files && argc > 5) {
            // TransformToShow + normalize of main.cpp:98-110 reduced to its effect on a valid map: rejected
            // pixels (inf) black, disparities scaled to 0..255 by the range
            std::vector<unsigned char> show(n);
            for (size_t k = 0; k < n; k++) {
                const float d = leftDisp[k];
                show[k] = (d >= 0 && d < (float)dispRange) ? (unsigned char)(d * 255.0f / (float)(dispRange > 1 ? dispRange - 1 : 1) + 0.5f) : 0;
            }
            smt::imwrite(argv[5], show.data(), row, col, 1);
        }
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - g_hash_ms;
        fprintf(stderr, "host-buffer pipeline (PCIe-inclusive) %.2f ms for %dx%d D=%d\n", ms, col, row, dispRange);
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
