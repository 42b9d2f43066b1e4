// Counterpart of CBLSM/CBLSM.cpp:13-213 on a synthetic pair, the active lines in the file's own order:
//   :64-67   ArmLength{L,R,Up,Down}(imageL, tao = 25, ..., maxLength = 34, secLength = 17)   four by-value-tao calls
//   :101-104 the same on imageR
//   :133-134 ComputeAD / ComputeADRight (uchar)
//   :146     costAggregationV5(right volume, RIGHT arms)
//   :147     costAggregationV5(left volume, LEFT arms)
//   :149     costAggregationV5(left result again, LEFT arms)
//   :150     costAggregationV5(right result again, LEFT arms)            <- the left image's arms, as written
//   :152-153 ComputeDispOringin x 2
// (medianBlur :24-25 feeds nothing; copyMakeBorder :124-129 feeds only commented-out calls; LeftRightConsistency
// :160 and the display conversion are commented out / app shell.)  Host buffers in and out, everything computed by
// libsmt_hip.so through smt_host.hpp.  Prints FNV-1a hashes for tests/test_cpp_host_gpu.py.
//   usage: cblsm_main H W D seed
#include <cstdio>
#include <cstdlib>
#include "smt_host.hpp"

static uint64_t fnv(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t k = 0; k < n; k++) { h ^= b[k]; h *= 1099511628211ull; }
    return h;
}
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
    const int row = argc > 1 ? atoi(argv[1]) : 375, col = argc > 2 ? atoi(argv[2]) : 450;
    const int dispRange = argc > 3 ? atoi(argv[3]) : 60;                       // CBLSM.cpp:29
    const uint32_t seed = argc > 4 ? (uint32_t)atoi(argv[4]) : 6;
    try {
        using namespace smt;
        Image imageL, imageR;
        imageL.rows = imageR.rows = row; imageL.cols = imageR.cols = col; imageL.channels = imageR.channels = 1;
        synth(row, col, dispRange, seed, imageL.data, imageR.data);
        const int winSize = 1;                                                  // :28
        const unsigned char tao = 25;                                           // :30
        const int maxLength = 34, secLength = 17;                               // :31-32
        const size_t n = (size_t)row * col, V = n * dispRange;
        std::vector<float> dispVolumLeft(V), dispVolumRight(V), costVolumLeft(V), costVolumRight(V), costVolumLeftSec(V),
            costVolumRightSec(V), dispLeft(n), dispRight(n);
        std::vector<int> ArmLL(n), ArmLR(n), ArmLup(n), ArmLdown(n), ArmRL(n), ArmRR(n), ArmRup(n), ArmRdown(n);
        ArmLengthL(imageL, tao, ArmLL.data(), maxLength, secLength);            // :64-67
        ArmLengthR(imageL, tao, ArmLR.data(), maxLength, secLength);
        ArmLengthUp(imageL, tao, ArmLup.data(), maxLength, secLength);
        ArmLengthDown(imageL, tao, ArmLdown.data(), maxLength, secLength);
        unsigned char *leftPtr = imageL.data.data(), *rightPtr = imageR.data.data();   // :89-99
        ArmLengthL(imageR, tao, ArmRL.data(), maxLength, secLength);            // :101-104
        ArmLengthR(imageR, tao, ArmRR.data(), maxLength, secLength);
        ArmLengthUp(imageR, tao, ArmRup.data(), maxLength, secLength);
        ArmLengthDown(imageR, tao, ArmRdown.data(), maxLength, secLength);
        ComputeAD(col, row, dispRange, leftPtr, rightPtr, dispVolumLeft.data());        // :133
        ComputeADRight(col, row, dispRange, leftPtr, rightPtr, dispVolumRight.data());  // :134
        costAggregationV5(dispVolumRight.data(), costVolumRight.data(), ArmRL.data(), ArmRR.data(), ArmRup.data(), ArmRdown.data(), dispRange, row, col, winSize);   // :146
        costAggregationV5(dispVolumLeft.data(), costVolumLeft.data(), ArmLL.data(), ArmLR.data(), ArmLup.data(), ArmLdown.data(), dispRange, row, col, winSize);      // :147
        costAggregationV5(costVolumLeft.data(), costVolumLeftSec.data(), ArmLL.data(), ArmLR.data(), ArmLup.data(), ArmLdown.data(), dispRange, row, col, winSize);   // :149
        costAggregationV5(costVolumRight.data(), costVolumRightSec.data(), ArmLL.data(), ArmLR.data(), ArmLup.data(), ArmLdown.data(), dispRange, row, col, winSize); // :150
        ComputeDispOringin(costVolumLeftSec.data(), dispLeft.data(), dispRange, row, col);     // :152
        ComputeDispOringin(costVolumRightSec.data(), dispRight.data(), dispRange, row, col);   // :153
        const char *an[8] = {"arm_LL", "arm_LR", "arm_Lup", "arm_Ldown", "arm_RL", "arm_RR", "arm_Rup", "arm_Rdown"};
        const std::vector<int> *av[8] = {&ArmLL, &ArmLR, &ArmLup, &ArmLdown, &ArmRL, &ArmRR, &ArmRup, &ArmRdown};
        for (int k = 0; k < 8; k++) printf("%s %016llx\n", an[k], (unsigned long long)fnv(av[k]->data(), n * 4));
        printf("ad_left %016llx\nad_right %016llx\n", (unsigned long long)fnv(dispVolumLeft.data(), V * 4),
               (unsigned long long)fnv(dispVolumRight.data(), V * 4));
        printf("agg_left %016llx\nagg_right %016llx\n", (unsigned long long)fnv(costVolumLeft.data(), V * 4),
               (unsigned long long)fnv(costVolumRight.data(), V * 4));
        printf("agg_left_sec %016llx\nagg_right_sec %016llx\n", (unsigned long long)fnv(costVolumLeftSec.data(), V * 4),
               (unsigned long long)fnv(costVolumRightSec.data(), V * 4));
        printf("disp_left %016llx\ndisp_right %016llx\n", (unsigned long long)fnv(dispLeft.data(), n * 4),
               (unsigned long long)fnv(dispRight.data(), n * 4));
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
