// Counterpart of SAD/SADmain.cpp, NCC/NCC_main.cpp and ASW/ASWeight.cpp on a synthetic pair: host
// buffers in, host maps out, everything computed by libsmt_hip.so through smt_host.hpp.  Prints
// FNV-1a hashes for tests/test_cpp_host_gpu.py.   usage: matchers_main H W D seed
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
static std::vector<unsigned char> pad(const std::vector<unsigned char> &a, int H, int W, int p)   // copyMakeBorder REPLICATE
{
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    std::vector<unsigned char> o((size_t)Hp * Wp);
    for (int i = 0; i < Hp; i++)
        for (int j = 0; j < Wp; j++) {
            int ii = i - p, jj = j - p;
            ii = ii < 0 ? 0 : ii > H - 1 ? H - 1 : ii;
            jj = jj < 0 ? 0 : jj > W - 1 ? W - 1 : jj;
            o[(size_t)i * Wp + j] = a[(size_t)ii * W + jj];
        }
    return o;
}

int main(int argc, char **argv)
{
    const int H = argc > 1 ? atoi(argv[1]) : 40, W = argc > 2 ? atoi(argv[2]) : 90, D = argc > 3 ? atoi(argv[3]) : 32;
    const uint32_t seed = argc > 4 ? (uint32_t)atoi(argv[4]) : 5;
    try {
        std::vector<unsigned char> L, R;
        synth(H, W, D, seed, L, R);
        const size_t n = (size_t)H * W;
        {   // SADmain.cpp:33-34,47-61 (winsize = 3 -> 9x9)
            const int winsize = 3, w = winsize + 1;
            auto Lp = pad(L, H, W, w), Rp = pad(R, H, W, w);
            std::vector<int> dl(n, 0), dr(n, 0);
            smt::GetPointDepthLeft(dl.data(), Lp.data(), Rp.data(), H + 2 * w, W + 2 * w, D, winsize);
            smt::GetPointDepthRight(dr.data(), Lp.data(), Rp.data(), H + 2 * w, W + 2 * w, D, winsize);
            printf("sad_left %016llx\nsad_right %016llx\n", (unsigned long long)fnv(dl.data(), n * 4),
                   (unsigned long long)fnv(dr.data(), n * 4));
        }
        {   // NCC_main.cpp:17-33 (smaller window here)
            std::vector<int> d(n, 0);
            smt::NCC_algorithem(L.data(), R.data(), W, H, d.data(), 3, D);
            printf("ncc %016llx\n", (unsigned long long)fnv(d.data(), n * 4));
        }
        {   // ASWeight.cpp:43-61
            const int winSize = 3, w = winSize + 1, T = 40;
            std::vector<double> sp, cm;
            smt::getMasks(sp, cm, winSize, 50, 30);
            auto Lp = pad(L, H, W, w), Rp = pad(R, H, W, w);
            std::vector<float> dl(n), dr(n);
            smt::AdaptiveSupportWeight(dl.data(), Lp.data(), Rp.data(), H + 2 * w, W + 2 * w, winSize, D, sp, cm, T, true);
            smt::AdaptiveSupportWeight(dr.data(), Lp.data(), Rp.data(), H + 2 * w, W + 2 * w, winSize, D, sp, cm, T, false);
            printf("asw_left %016llx\nasw_right %016llx\n", (unsigned long long)fnv(dl.data(), n * 4),
                   (unsigned long long)fnv(dr.data(), n * 4));
            std::vector<float> med(n);
            smt::MedianFilter(dl.data(), med.data(), W, H, 3);             // main.cpp:94
            printf("median %016llx\n", (unsigned long long)fnv(med.data(), n * 4));
            smt::RemoveSpeckles(dl.data(), W, H, 1, 30, -2147483647 - 1);   // main.cpp:93 with int(+inf) on x86
            printf("speckles %016llx\n", (unsigned long long)fnv(dl.data(), n * 4));
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
