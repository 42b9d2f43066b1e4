// C entry point around the REFERENCE's own CrossAggregator class
// (/root/reference/CBLSM/cross_aggregator.{h,cpp}, compiled unmodified from where it
// lies -- see oracle/Makefile).  This wrapper contains no algorithm: it only drives the
// reference's public API in the order CBLSM.cpp:138-143 does.  Test infrastructure.
#include "cross_aggregator.h"
#include <cstring>

extern "C" __attribute__((visibility("default")))
int ref_crossagg(const unsigned char* bgr_left, const float* cost_init, int width, int height,
                 int disp_range, int L1, int L2, int t1, int t2, int iters,
                 unsigned char* arms_out /* [N][4] left,right,top,bottom */, float* cost_out)
{
    CrossAggregator agg;
    if (!agg.Initialize(width, height, 0, disp_range)) return 1;
    agg.SetData(bgr_left, bgr_left, cost_init);
    agg.SetParams(L1, L2, t1, t2);
    agg.Aggregate(iters);
    const CrossArm* arms = agg.get_arms_ptr();
    const size_t n = size_t(width) * height;
    for (size_t p = 0; p < n; ++p) {
        arms_out[4 * p + 0] = arms[p].left;  arms_out[4 * p + 1] = arms[p].right;
        arms_out[4 * p + 2] = arms[p].top;   arms_out[4 * p + 3] = arms[p].bottom;
    }
    std::memcpy(cost_out, agg.get_cost_ptr(), n * disp_range * sizeof(float));
    return 0;
}
