// smt_host.hpp -- C++ host-side mirror of the reference's classes over the C ABI (smt.h).
//
// Same class names, method names, argument order and HOST-pointer ownership as the
// reference (AD-CensusV1/{AD-Census.h,CrossArm.h,ScanlineOptimizer.h,PostProcessing.h},
// CBLSM/cross_aggregator.h), so that the reference's main()s compile against this header
// instead of their own after dropping the cv::Mat arguments.  Every
// method uploads / downloads through smt_memcpy_* and runs the HIP kernels of
// libsmt_hip.so; there is no CPU implementation behind it.
//
// Deviations (all forced by the missing OpenCV dependency, none changes results):
//   - cv::Mat parameters become (const unsigned char* data, int channels).
//   - Errors: the reference returns void and has UB on bad input; these methods throw
//     std::runtime_error carrying smt_strerror().
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "../../include/smt.h"

namespace smt {

inline void check(int rc, const char *what)
{
    if (rc != SMT_OK) throw std::runtime_error(std::string(what) + ": " + smt_strerror(rc));
}

// RAII device buffer
template <class T> class DevBuf {
public:
    DevBuf() = default;
    explicit DevBuf(size_t n) { resize(n); }
    ~DevBuf() { if (p_) smt_free(p_); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    void resize(size_t n)
    {
        if (p_) smt_free(p_);
        p_ = nullptr; n_ = n;
        void *q = nullptr;
        check(smt_malloc(&q, n * sizeof(T)), "smt_malloc");
        p_ = static_cast<T *>(q);
    }
    void upload(const T *h) { check(smt_memcpy_h2d(p_, h, n_ * sizeof(T), nullptr), "h2d"); }
    void download(T *h) const
    {
        check(smt_memcpy_d2h(h, p_, n_ * sizeof(T), nullptr), "d2h");
        check(smt_stream_sync(nullptr), "sync");
    }
    T *get() const { return p_; }
    size_t size() const { return n_; }
private:
    T *p_ = nullptr;
    size_t n_ = 0;
};

// ------------------------------------------------------------------ AD-Census.h:9-43
class AD_Census {
public:
    AD_Census() = default;
    ~AD_Census() { if (h_) smt_adcensus_destroy(h_); }
    // Initialize(leftImage, rightImage, dispRange, row, col, LImage, RImage, sigmaC, sigmaS)
    void Initialize(float *leftImage, float *rightImage, int dispRange, int row, int col, float sigmaC,
                    float sigmaS)
    {
        row_ = row; col_ = col; D_ = dispRange;
        if (h_) { smt_adcensus_destroy(h_); h_ = nullptr; }
        check(smt_adcensus_create(row, col, dispRange, sigmaC, sigmaS, &h_), "smt_adcensus_create");
        L_.resize((size_t)row * col); R_.resize((size_t)row * col);
        L_.upload(leftImage); R_.upload(rightImage);
        hostL_.assign((size_t)row * col * dispRange, 0.f);
        hostR_.assign((size_t)row * col * dispRange, 0.f);
        validL_ = validR_ = false;
    }
    void ComputeADcensus() { check(smt_adcensus_compute(h_, L_.get(), R_.get(), SMT_VIEW_LEFT, nullptr, nullptr), "compute"); validL_ = false; }
    void ComputeADcensusRight() { check(smt_adcensus_compute(h_, L_.get(), R_.get(), SMT_VIEW_RIGHT, nullptr, nullptr), "compute"); validR_ = false; }
    // device-side volume (borrowed), for chaining into CrossArmAggregation without a round trip
    float *DevicePtrLeft() { float *p; check(smt_adcensus_volume(h_, SMT_VIEW_LEFT, &p), "volume"); return p; }
    float *DevicePtrRight() { float *p; check(smt_adcensus_volume(h_, SMT_VIEW_RIGHT, &p), "volume"); return p; }
    // GetPtrLeft/Right: host copy of the cost volume, valid for the object's life like the reference's
    // (bad input -- a pixel that is not an integer in 0..255 -- is reported here, at the first point
    // where the reference's caller would consume the volume: smt_adcensus_status)
    float *GetPtrLeft()
    {
        if (!validL_) { check(smt_adcensus_status(h_), "smt_adcensus_status"); fetch(DevicePtrLeft(), hostL_); validL_ = true; }
        return hostL_.data();
    }
    float *GetPtrRight()
    {
        if (!validR_) { check(smt_adcensus_status(h_), "smt_adcensus_status"); fetch(DevicePtrRight(), hostR_); validR_ = true; }
        return hostR_.data();
    }
    void WTA(float *leftdisp, float *rightDisp)
    {
        const size_t n = (size_t)row_ * col_;
        DevBuf<float> dl(n), dr(n);
        check(smt_wta(DevicePtrLeft(), row_, col_, D_, dl.get(), nullptr), "wta");
        check(smt_wta(DevicePtrRight(), row_, col_, D_, dr.get(), nullptr), "wta");
        dl.download(leftdisp); dr.download(rightDisp);
        check(smt_adcensus_status(h_), "smt_adcensus_status");
    }
private:
    void fetch(const float *dev, std::vector<float> &host)
    {
        check(smt_memcpy_d2h(host.data(), dev, host.size() * sizeof(float), nullptr), "d2h");
        check(smt_stream_sync(nullptr), "sync");
    }
    smt_adcensus *h_ = nullptr;
    int row_ = 0, col_ = 0, D_ = 0;
    DevBuf<float> L_, R_;
    std::vector<float> hostL_, hostR_;
    bool validL_ = false, validR_ = false;
};

// ------------------------------------------------------------------ CrossArm.h:9-36
class CrossArmAggregation {
public:
    CrossArmAggregation() = default;
    ~CrossArmAggregation() { if (h_) smt_crossarm_destroy(h_); }
    void Initialize(int row, int col, float * /*leftImage*/, float * /*rightImage*/, int tao, int dispRange)
    {
        row_ = row; col_ = col; D_ = dispRange;
        if (h_) { smt_crossarm_destroy(h_); h_ = nullptr; }
        smt_crossarm_params p;
        smt_crossarm_default_params(&p);
        p.tau = tao;
        check(smt_crossarm_create(row, col, dispRange, &p, &h_), "smt_crossarm_create");
        started_ = false;
    }
    // The four Compute*ArmLength(const Mat&) calls of main.cpp:69-72 always come together and in
    // this order (the threshold state chains through them), so they are one call here.
    void ComputeArmLengths(const unsigned char *image, int channels)
    {
        DevBuf<unsigned char> img((size_t)row_ * col_ * channels);
        img.upload(image);
        check(smt_crossarm_arms(h_, img.get(), channels), "smt_crossarm_arms");
        check(smt_stream_sync(nullptr), "sync");
    }
    // The reference's four calls one by one (CrossArm.h:15-18), for callers that run a subset or another
    // order; the sticky threshold is whatever the previous call left.  Initialize resets it.
    void ComputeLeftArmLength(const unsigned char *image, int channels) { arm_dir(image, channels, 0); }
    void ComputeRightArmLength(const unsigned char *image, int channels) { arm_dir(image, channels, 1); }
    void ComputeTopArmLength(const unsigned char *image, int channels) { arm_dir(image, channels, 2); }
    void ComputeButtonArmLength(const unsigned char *image, int channels) { arm_dir(image, channels, 3); }
    // CrossArm.cpp:104-145 (declared in CrossArm.h:19, never called): exclusive upper bounds
    void Aggregation(float *dispVolume, float *aggregatedCostVolume)
    {
        const size_t V = (size_t)row_ * col_ * D_;
        DevBuf<float> in(V), out(V);
        in.upload(dispVolume);
        check(smt_crossarm_aggregate(h_, in.get(), out.get(), 2, nullptr), "smt_crossarm_aggregate");
        out.download(aggregatedCostVolume);
        check(smt_crossarm_status(h_), "smt_crossarm_status");
    }
    void AggregationVertical(float *dispVolume, float *aggregatedCostVolume)
    {
        const size_t V = (size_t)row_ * col_ * D_;
        DevBuf<float> in(V), out(V);
        in.upload(dispVolume);
        AggregationVerticalDevice(in.get(), out.get());
        out.download(aggregatedCostVolume);
        check(smt_crossarm_status(h_), "smt_crossarm_status");
    }
    void AggregationVerticalDevice(const float *dev_in, float *dev_out, float *dev_disp = nullptr)
    {
        check(smt_crossarm_aggregate(h_, dev_in, dev_out, 0, dev_disp), "smt_crossarm_aggregate");
    }
    void WTA(float *AggredCostVolume, float *disp)
    {
        const size_t n = (size_t)row_ * col_;
        DevBuf<float> v(n * D_), d(n);
        v.upload(AggredCostVolume);
        check(smt_wta(v.get(), row_, col_, D_, d.get(), nullptr), "wta");
        d.download(disp);
    }
private:
    void arm_dir(const unsigned char *image, int channels, int dir)
    {
        DevBuf<unsigned char> img((size_t)row_ * col_ * channels);
        img.upload(image);
        if (!started_) { check(smt_crossarm_reset(h_), "smt_crossarm_reset"); started_ = true; }
        check(smt_crossarm_arm_dir(h_, img.get(), channels, dir), "smt_crossarm_arm_dir");
        check(smt_stream_sync(nullptr), "sync");
    }
    smt_crossarm *h_ = nullptr;
    int row_ = 0, col_ = 0, D_ = 0;
    bool started_ = false;
};

// ------------------------------------------------------------------ ScanlineOptimizer.h:8-34
class ScanlineOptimizer {
public:
    ScanlineOptimizer() = default;
    ~ScanlineOptimizer() { if (h_) smt_scanline_destroy(h_); }
    void Initialize(int row, int col, int dispRange, float * /*costVolume*/, int p1, int p2)
    {
        row_ = row; col_ = col; D_ = dispRange;
        if (h_) { smt_scanline_destroy(h_); h_ = nullptr; }
        check(smt_scanline_create(row, col, dispRange, p1, p2, &h_), "smt_scanline_create");
        processed_.resize((size_t)row * col * dispRange);
    }
    void ScanLine(float *costVolume, float *Image)
    {
        const size_t n = (size_t)row_ * col_;
        DevBuf<float> in(n * D_), g(n);
        in.upload(costVolume); g.upload(Image);
        check(smt_scanline_run(h_, in.get(), g.get(), processed_.get(), nullptr), "smt_scanline_run");
        check(smt_stream_sync(nullptr), "sync");
    }
    void WTA(float *disp)
    {
        DevBuf<float> d((size_t)row_ * col_);
        check(smt_wta(processed_.get(), row_, col_, D_, d.get(), nullptr), "wta");
        d.download(disp);
    }
    float *DeviceProcessedVolume() { return processed_.get(); }
private:
    smt_scanline *h_ = nullptr;
    int row_ = 0, col_ = 0, D_ = 0;
    DevBuf<float> processed_;
};

// ------------------------------------------------------------------ PostProcessing.h:72
inline void LeftRightConsistency(int col, int row, int gate, float *leftDisp, float *rightDisp,
                                 std::vector<std::pair<int, int>> &occlusion,
                                 std::vector<std::pair<int, int>> &mismatch)
{
    const size_t n = (size_t)row * col;
    DevBuf<float> dl(n), dr(n);
    DevBuf<uint8_t> cls(n);
    dl.upload(leftDisp); dr.upload(rightDisp);
    check(smt_lrcheck(dl.get(), dr.get(), row, col, gate, cls.get(), nullptr, nullptr), "smt_lrcheck");
    dl.download(leftDisp);
    std::vector<uint8_t> h(n);
    cls.download(h.data());
    std::vector<int> o(2 * n), m(2 * n);
    int no = 0, nm = 0;
    check(smt_lrcheck_lists(h.data(), row, col, o.data(), &no, m.data(), &nm), "smt_lrcheck_lists");
    occlusion.clear(); mismatch.clear();
    for (int k = 0; k < no; k++) occlusion.emplace_back(o[2 * k], o[2 * k + 1]);
    for (int k = 0; k < nm; k++) mismatch.emplace_back(m[2 * k], m[2 * k + 1]);
}

// ------------------------------------------------------------------ PostProcessing.h:10 (no call site)
inline void LeftAndRightConsistency(float *leftDisp, float *rightDisp, float *lastDisp, int col, int row, float gate,
                                    std::vector<std::pair<int, int>> &occlusion,
                                    std::vector<std::pair<int, int>> &mismatch)
{
    const size_t n = (size_t)row * col;
    DevBuf<float> dl(n), dr(n), last(n);
    DevBuf<uint8_t> cls(n);
    dl.upload(leftDisp); dr.upload(rightDisp);
    check(smt_lrcheck_variant(dl.get(), dr.get(), last.get(), row, col, gate, cls.get(), nullptr, nullptr), "smt_lrcheck_variant");
    last.download(lastDisp);
    std::vector<uint8_t> h(n);
    cls.download(h.data());
    std::vector<int> o(2 * n), m(2 * n);
    int no = 0, nm = 0;
    check(smt_lrcheck_lists(h.data(), row, col, o.data(), &no, m.data(), &nm), "smt_lrcheck_lists");
    for (int k = 0; k < no; k++) occlusion.emplace_back(o[2 * k], o[2 * k + 1]);     // the reference appends (:44, :47)
    for (int k = 0; k < nm; k++) mismatch.emplace_back(m[2 * k], m[2 * k + 1]);
}

// ------------------------------------------------------------------ PostProcessing.h:156
// Same signature and side effects as the reference, the replaced `mismatch` list included (:186).
inline void FillTheHole(const int row, const int col, const int dispRange, float *dispLeft,
                        std::vector<std::pair<int, int>> &occlusion, std::vector<std::pair<int, int>> &mismatch)
{
    const size_t n = (size_t)row * col;
    DevBuf<float> d(n);
    d.upload(dispLeft);
    std::vector<int> o(2 * occlusion.size()), m(2 * mismatch.size()), third(2 * n);
    for (size_t k = 0; k < occlusion.size(); k++) { o[2 * k] = occlusion[k].first; o[2 * k + 1] = occlusion[k].second; }
    for (size_t k = 0; k < mismatch.size(); k++) { m[2 * k] = mismatch[k].first; m[2 * k + 1] = mismatch[k].second; }
    int nt = -1;
    check(smt_fill_the_hole(d.get(), row, col, dispRange, o.data(), (int)occlusion.size(), m.data(),
                            (int)mismatch.size(), third.data(), &nt, nullptr), "smt_fill_the_hole");
    d.download(dispLeft);
    if (nt >= 0) {
        mismatch.clear();
        for (int k = 0; k < nt; k++) mismatch.emplace_back(third[2 * k], third[2 * k + 1]);
    }
}

// ------------------------------------------------------------------ CBLSM.h:65-236
// chooseArmLength{Left,Right,Up,Down} on host arm maps (call sites commented out in CBLSM.cpp:108-111)
inline void choose_arm_length_(int dir, const int *own, const int *vert, const int *RL, const int *RR, int dispRange,
                               int *Armvolume, int row, int col)
{
    const size_t n = (size_t)row * col;
    DevBuf<int> a(n), v(n), l(n), r(n), out(n * dispRange);
    a.upload(own); l.upload(RL); r.upload(RR);
    if (vert) v.upload(vert);
    check(smt_cblsm_choose_arm_length(dir, a.get(), vert ? v.get() : nullptr, l.get(), r.get(), row, col, dispRange,
                                      out.get(), nullptr), "smt_cblsm_choose_arm_length");
    out.download(Armvolume);
}
inline void chooseArmLengthLeft(int *ArmLL, int * /*ArmLR*/, int *ArmRL, int *ArmRR, int dispRange, int *Armvolume, int row, int col)
{ choose_arm_length_(0, ArmLL, nullptr, ArmRL, ArmRR, dispRange, Armvolume, row, col); }
inline void chooseArmLengthRight(int * /*ArmLL*/, int *ArmLR, int *ArmRL, int *ArmRR, int dispRange, int *Armvolume, int row, int col)
{ choose_arm_length_(1, ArmLR, nullptr, ArmRL, ArmRR, dispRange, Armvolume, row, col); }
inline void chooseArmLengthUp(int *ArmLUp, int * /*ArmLDown*/, int *ArmRUp, int * /*ArmRDown*/, int *ArmRL, int *ArmRR, int dispRange,
                              int *Armvolume, int row, int col)
{ choose_arm_length_(2, ArmLUp, ArmRUp, ArmRL, ArmRR, dispRange, Armvolume, row, col); }
inline void chooseArmLengthDown(int * /*ArmLUp*/, int *ArmLDown, int * /*ArmRUp*/, int *ArmRDown, int *ArmRL, int *ArmRR, int dispRange,
                                int *Armvolume, int row, int col)
{ choose_arm_length_(3, ArmLDown, ArmRDown, ArmRL, ArmRR, dispRange, Armvolume, row, col); }

// ------------------------------------------------------------------ cross_aggregator.h:27-113
struct CrossArm { uint8_t left, right, top, bottom; };

class CrossAggregator {
public:
    CrossAggregator() = default;
    ~CrossAggregator() { if (h_) smt_crossagg_destroy(h_); }
    bool Initialize(const int &width, const int &height, const int &min_disparity, const int &max_disparity)
    {
        w_ = width; hgt_ = height; D_ = max_disparity - min_disparity;
        if (h_) { smt_crossagg_destroy(h_); h_ = nullptr; }
        const int rc = smt_crossagg_create(width, height, D_, &h_);
        if (rc == SMT_ERR_ARG) return false;            // reference returns false, :28-31
        check(rc, "smt_crossagg_create");
        return true;
    }
    void SetData(const uint8_t *img_left, const uint8_t * /*img_right*/, const float *cost_init)
    {
        img_ = img_left; cost_ = cost_init;
    }
    void SetParams(const int &L1, const int &L2, const int &t1, const int &t2)
    {
        if (h_) check(smt_crossagg_set_params(h_, L1, L2, t1, t2), "smt_crossagg_set_params");
    }
    void Aggregate(const int &num_iters)
    {
        if (!h_) return;                                 // :91-93
        const size_t n = (size_t)w_ * hgt_;
        DevBuf<uint8_t> img(n * 3);
        DevBuf<float> c(n * D_);
        img.upload(img_); c.upload(cost_);
        check(smt_crossagg_aggregate(h_, img.get(), c.get(), num_iters), "smt_crossagg_aggregate");
        host_cost_.resize(n * D_);
        host_arms_.resize(n);
        float *dc; uint8_t *da;
        check(smt_crossagg_cost(h_, &dc), "cost"); check(smt_crossagg_arms(h_, &da), "arms");
        check(smt_memcpy_d2h(host_cost_.data(), dc, n * D_ * 4, nullptr), "d2h");
        check(smt_memcpy_d2h(host_arms_.data(), da, n * 4, nullptr), "d2h");
        check(smt_stream_sync(nullptr), "sync");
    }
    CrossArm *get_arms_ptr() { return host_arms_.data(); }
    float *get_cost_ptr() { return host_cost_.empty() ? nullptr : host_cost_.data(); }
private:
    smt_crossagg *h_ = nullptr;
    int w_ = 0, hgt_ = 0, D_ = 0;
    const uint8_t *img_ = nullptr;
    const float *cost_ = nullptr;
    std::vector<float> host_cost_;
    std::vector<CrossArm> host_arms_;
};

// ------------------------------------------------------------------ window matchers (host buffers)
// Images are the caller's replicate-padded uint8 buffers, exactly what the reference's drivers pass
// after copyMakeBorder (SADmain.cpp:47-48, ASWeight.cpp:54-57); rows/cols are the PADDED sizes like
// Mat::rows/cols in the reference signatures.

// GetPointDepthLeft / GetPointDepthRight (Sad.h:96-182): disparity is int32 [rows-2w][cols-2w], w = winsize+1
inline void GetPointDepth(int *disparity, const unsigned char *leftimg, const unsigned char *rightimg, int rows,
                          int cols, int MaxDisparity, int winsize, bool left_view)
{
    const int w = winsize + 1, H = rows - 2 * w, W = cols - 2 * w;
    DevBuf<unsigned char> L((size_t)rows * cols), R((size_t)rows * cols);
    DevBuf<int> d((size_t)H * W);
    L.upload(leftimg); R.upload(rightimg);
    check(smt_sad(L.get(), R.get(), H, W, MaxDisparity, winsize, left_view ? SMT_VIEW_LEFT : SMT_VIEW_RIGHT, d.get(),
                  nullptr), "smt_sad");
    d.download(disparity);
}
inline void GetPointDepthLeft(int *disparity, const unsigned char *l, const unsigned char *r, int rows, int cols,
                              int MaxDisparity, int winsize) { GetPointDepth(disparity, l, r, rows, cols, MaxDisparity, winsize, true); }
inline void GetPointDepthRight(int *disparity, const unsigned char *l, const unsigned char *r, int rows, int cols,
                               int MaxDisparity, int winsize) { GetPointDepth(disparity, l, r, rows, cols, MaxDisparity, winsize, false); }

// NCC_algorithem(leftImage, rigthImage, width, height, disp, winSize, dispRange) (NCC.h:69-95)
inline void NCC_algorithem(const unsigned char *leftImage, const unsigned char *rigthImage, int width, int height,
                           int *disp, int winSize, int dispRange)
{
    const size_t n = (size_t)width * height;
    DevBuf<unsigned char> L(n), R(n);
    DevBuf<int> d(n);
    L.upload(leftImage); R.upload(rigthImage);
    check(smt_ncc(L.get(), R.get(), height, width, dispRange, winSize, d.get(), nullptr, nullptr), "smt_ncc");
    d.download(disp);
}

// getGausssianMask + getColorMask (ASW.h:16-47)
inline void getMasks(std::vector<double> &spaceMask, std::vector<double> &colorMask, int winSize, double spaceSigma,
                     double colorSigma)
{
    const int side = 2 * winSize + 3;
    spaceMask.assign((size_t)side * side, 0.0);
    colorMask.assign(256, 0.0);
    check(smt_asw_masks(winSize, spaceSigma, colorSigma, spaceMask.data(), colorMask.data()), "smt_asw_masks");
}

// AdaptiveSupportWeight / AdaptiveSupportWeightRight (ASW.h:329-431); disp float [rows-2w][cols-2w], w = winSize+1
inline void AdaptiveSupportWeight(float *disp, const unsigned char *leftGray, const unsigned char *rightGray, int rows,
                                  int cols, int winSize, int dispRange, const std::vector<double> &space,
                                  const std::vector<double> &color, int T, bool left_view = true)
{
    const int w = winSize + 1, H = rows - 2 * w, W = cols - 2 * w;
    DevBuf<unsigned char> L((size_t)rows * cols), R((size_t)rows * cols);
    DevBuf<double> sp(space.size()), cm(color.size());
    DevBuf<float> d((size_t)H * W);
    L.upload(leftGray); R.upload(rightGray); sp.upload(space.data()); cm.upload(color.data());
    check(smt_asw(L.get(), R.get(), H, W, dispRange, winSize, sp.get(), cm.get(), T,
                  left_view ? SMT_VIEW_LEFT : SMT_VIEW_RIGHT, d.get(), nullptr, nullptr), "smt_asw");
    d.download(disp);
}

// MedianFilter / RemoveSpeckles (PostProcessing.h:250-344) on host maps
inline void MedianFilter(const float *in, float *out, const int &width, const int &height, const int wnd_size)
{
    const size_t n = (size_t)width * height;
    DevBuf<float> a(n), b(n);
    a.upload(in);
    check(smt_median_filter(a.get(), b.get(), width, height, wnd_size, nullptr), "smt_median_filter");
    b.download(out);
}
inline void RemoveSpeckles(float *disparity_map, const int &width, const int &height, const int &diff_insame,
                           const unsigned int &min_speckle_aera, const int &invalid_val)
{
    const size_t n = (size_t)width * height;
    DevBuf<float> a(n);
    a.upload(disparity_map);
    check(smt_remove_speckles(a.get(), width, height, diff_insame, min_speckle_aera, invalid_val, nullptr),
          "smt_remove_speckles");
    a.download(disparity_map);
}

// ------------------------------------------------------------------ drivers' image handling
// imread(path) (AD-CensusV1/main.cpp:16-17): 3-channel B, G, R;  cvtColor(.., CV_BGR2GRAY) (:19-20);
// imwrite(path, img) (:115-117).  Host-side decoding / encoding is libsmt_hip.so's own (PNG, PGM, PPM).
struct Image {
    int rows = 0, cols = 0, channels = 0;
    std::vector<unsigned char> data;
};
inline Image imread(const std::string &path, int want_channels = 3)
{
    Image im;
    uint8_t *p = nullptr;
    check(smt_image_read(path.c_str(), want_channels, &p, &im.rows, &im.cols, &im.channels), "smt_image_read");
    im.data.assign(p, p + (size_t)im.rows * im.cols * im.channels);
    smt_image_free(p);
    return im;
}
inline void imwrite(const std::string &path, const unsigned char *data, int rows, int cols, int channels)
{
    check(smt_image_write(path.c_str(), data, rows, cols, channels), "smt_image_write");
}
inline Image cvtColorBGR2GRAY(const Image &bgr)
{
    if (bgr.channels != 3) throw std::runtime_error("cvtColorBGR2GRAY: 3-channel image expected");
    const size_t n = (size_t)bgr.rows * bgr.cols;
    DevBuf<unsigned char> in(n * 3), out(n);
    in.upload(bgr.data.data());
    check(smt_bgr2gray(in.get(), bgr.rows, bgr.cols, out.get(), nullptr), "smt_bgr2gray");
    Image g;
    g.rows = bgr.rows; g.cols = bgr.cols; g.channels = 1; g.data.resize(n);
    out.download(g.data.data());
    return g;
}

// ------------------------------------------------------------------ CBLSM.h active path (CBLSM.cpp:64-67, 101-104, 133-153)
// ArmLength{L,R,Up,Down}(const Mat&, uchar tao, int* arm, int maxLength, int secLength) (CBLSM.h:643, 536, 753, 861):
// `tao` is BY VALUE -- every call starts from it again, the drop to 6 past secLength is local to the call.
inline void arm_length_(const Image &image, unsigned char tao, int *arm, int maxLength, int secLength, int dir)
{
    smt_crossarm_params p;
    smt_crossarm_cblsm_params(&p);
    p.tau = tao; p.max_length = maxLength; p.sec_length = secLength;
    smt_crossarm *h = nullptr;
    check(smt_crossarm_create(image.rows, image.cols, 1, &p, &h), "smt_crossarm_create");
    try {
        DevBuf<unsigned char> img(image.data.size());
        img.upload(image.data.data());
        check(smt_crossarm_reset(h), "smt_crossarm_reset");
        check(smt_crossarm_arm_dir(h, img.get(), image.channels, dir), "smt_crossarm_arm_dir");
        int *maps[4] = {nullptr, nullptr, nullptr, nullptr};
        check(smt_crossarm_arm_maps(h, &maps[0], &maps[1], &maps[2], &maps[3]), "smt_crossarm_arm_maps");
        check(smt_memcpy_d2h(arm, maps[dir], (size_t)image.rows * image.cols * sizeof(int), nullptr), "d2h");
        check(smt_stream_sync(nullptr), "sync");
    } catch (...) { smt_crossarm_destroy(h); throw; }
    smt_crossarm_destroy(h);
}
inline void ArmLengthL(const Image &image, unsigned char tao, int *LArm, int maxLength, int secLength) { arm_length_(image, tao, LArm, maxLength, secLength, 0); }
inline void ArmLengthR(const Image &image, unsigned char tao, int *RArm, int maxLength, int secLength) { arm_length_(image, tao, RArm, maxLength, secLength, 1); }
inline void ArmLengthUp(const Image &image, unsigned char tao, int *UpArm, int maxLength, int secLength) { arm_length_(image, tao, UpArm, maxLength, secLength, 2); }
inline void ArmLengthDown(const Image &image, unsigned char tao, int *DownArm, int maxLength, int secLength) { arm_length_(image, tao, DownArm, maxLength, secLength, 3); }

// ComputeAD / ComputeADRight (CBLSM.h:327-381): uchar absolute differences into a float [row][col][dispRange] volume
inline void compute_ad_(int col, int row, int dispRange, const unsigned char *leftImage, const unsigned char *rightImage,
                        float *ADcostVolum, int view)
{
    const size_t n = (size_t)row * col;
    DevBuf<unsigned char> L(n), R(n);
    DevBuf<float> vol(n * dispRange);
    L.upload(leftImage); R.upload(rightImage);
    check(smt_cblsm_ad(L.get(), R.get(), row, col, dispRange, view, vol.get(), nullptr), "smt_cblsm_ad");
    vol.download(ADcostVolum);
}
inline void ComputeAD(int col, int row, int dispRange, unsigned char *leftImage, unsigned char *rightImage, float *ADcostVolum)
{ compute_ad_(col, row, dispRange, leftImage, rightImage, ADcostVolum, SMT_VIEW_LEFT); }
inline void ComputeADRight(int col, int row, int dispRange, unsigned char *leftImage, unsigned char *rightImage, float *ADcostVolum)
{ compute_ad_(col, row, dispRange, leftImage, rightImage, ADcostVolum, SMT_VIEW_RIGHT); }

// costAggregationV5 (CBLSM.h:1179-1224): rectangle mean, rows outer / columns inner, with the four arm arrays the
// caller passes -- whichever image they were computed on (CBLSM.cpp:150 aggregates the right volume with the left
// image's arms).  winSize is unused by the reference too.
inline void costAggregationV5(float *dispvolume, float *CostVolume, int *ArmvolumeL, int *ArmvolumeR, int *ArmvolumeUp,
                              int *ArmvolumeDown, int dispRange, int row, int col, int /*winSize*/)
{
    const size_t n = (size_t)row * col;
    DevBuf<float> in(n * dispRange), out(n * dispRange);
    DevBuf<int> aL(n), aR(n), aU(n), aD(n);
    in.upload(dispvolume);
    aL.upload(ArmvolumeL); aR.upload(ArmvolumeR); aU.upload(ArmvolumeUp); aD.upload(ArmvolumeDown);
    smt_crossarm_params p;
    smt_crossarm_cblsm_params(&p);
    smt_crossarm *h = nullptr;
    check(smt_crossarm_create(row, col, dispRange, &p, &h), "smt_crossarm_create");
    try {
        check(smt_crossarm_load_arm_maps(h, aL.get(), aR.get(), aU.get(), aD.get()), "smt_crossarm_load_arm_maps");
        check(smt_crossarm_aggregate(h, in.get(), out.get(), 1, nullptr), "smt_crossarm_aggregate");
        out.download(CostVolume);
        check(smt_crossarm_status(h), "smt_crossarm_status");
    } catch (...) { smt_crossarm_destroy(h); throw; }
    smt_crossarm_destroy(h);
}

// ComputeDispOringin (CBLSM.h:383-407): first strict minimum over d
inline void ComputeDispOringin(float *costVolume, float *disp, int dispRange, int row, int col)
{
    const size_t n = (size_t)row * col;
    DevBuf<float> v(n * dispRange), d(n);
    v.upload(costVolume);
    check(smt_wta(v.get(), row, col, dispRange, d.get(), nullptr), "smt_wta");
    d.download(disp);
}

// ------------------------------------------------------------------ config 5: a batch over the node's GPUs
// One host thread, one smt_adcensus handle per device (smt_adcensus_create_on); pair b goes to device
// b % G as a contiguous block per device; no data-path exchange between devices (pairs are independent,
// SURVEY 8e) -- the maps come back to the host buffers directly.  L, R: [pairs][row][col] float (integer
// valued); dispL, dispR: [pairs][row][col] float out.  Returns the number of devices used.
inline int AD_Census_batch_all_devices(const float *L, const float *R, int pairs, int dispRange, int row, int col,
                                       float sigmaC, float sigmaS, float *dispL, float *dispR, int max_devices = 0)
{
    int G = 0;
    check(smt_device_count(&G), "smt_device_count");
    if (max_devices > 0 && G > max_devices) G = max_devices;
    if (G > pairs) G = pairs;
    if (G <= 0) throw std::runtime_error("no device / empty batch");
    const size_t n = (size_t)row * col;
    struct Dev { smt_adcensus *h = nullptr; void *stream = nullptr; float *L = nullptr, *R = nullptr, *dl = nullptr, *dr = nullptr; int start = 0, count = 0; };
    std::vector<Dev> dev(G);
    const int q = pairs / G, r = pairs % G;
    for (int g = 0; g < G; g++) {
        Dev &d = dev[g];
        d.count = q + (g < r ? 1 : 0);
        d.start = g * q + (g < r ? g : r);
        check(smt_set_device(g), "smt_set_device");
        check(smt_adcensus_create_on(g, row, col, dispRange, sigmaC, sigmaS, &d.h), "smt_adcensus_create_on");
        check(smt_stream_create(&d.stream), "smt_stream_create");
        check(smt_adcensus_set_stream(d.h, d.stream), "smt_adcensus_set_stream");
        const size_t bytes = (size_t)d.count * n * sizeof(float);
        check(smt_malloc((void **)&d.L, bytes), "smt_malloc"); check(smt_malloc((void **)&d.R, bytes), "smt_malloc");
        check(smt_malloc((void **)&d.dl, bytes), "smt_malloc"); check(smt_malloc((void **)&d.dr, bytes), "smt_malloc");
        check(smt_memcpy_h2d(d.L, L + (size_t)d.start * n, bytes, d.stream), "h2d");
        check(smt_memcpy_h2d(d.R, R + (size_t)d.start * n, bytes, d.stream), "h2d");
        // asynchronous on the device's own stream: all devices compute concurrently
        check(smt_adcensus_compute_batch(d.h, d.L, d.R, d.count, SMT_VIEW_BOTH, d.dl, d.dr), "smt_adcensus_compute_batch");
        check(smt_memcpy_d2h(dispL + (size_t)d.start * n, d.dl, bytes, d.stream), "d2h");
        check(smt_memcpy_d2h(dispR + (size_t)d.start * n, d.dr, bytes, d.stream), "d2h");
    }
    for (int g = 0; g < G; g++) {
        Dev &d = dev[g];
        check(smt_set_device(g), "smt_set_device");
        check(smt_adcensus_status(d.h), "smt_adcensus_status");        // synchronises the device's stream
        smt_free(d.L); smt_free(d.R); smt_free(d.dl); smt_free(d.dr);
        smt_adcensus_destroy(d.h);
        smt_stream_destroy(d.stream);
    }
    check(smt_set_device(0), "smt_set_device");
    return G;
}

#ifdef SMT_HOST_WITH_RCCL
}  // namespace smt
#include <rccl/rccl.h>
namespace smt {
// ------------------------------------------------------------------ config 5 with the exchange over RCCL / xGMI
// Same sharding as AD_Census_batch_all_devices, but the disparity maps stay on the devices and are exchanged the
// way SURVEY 8e prescribes: one ncclAllGather of the (padded) per-device [count][row][col] maps -- every device
// ends up with all `pairs` maps -- and one ncclAllReduce(sum) of a float64 checksum.  Single process, one
// communicator per device (ncclCommInitAll), group calls.  Returns the all-reduced checksum; gathered maps are
// copied back from device 0.  Pairs are independent, so this is the only traffic between devices.
inline void rccl_check(ncclResult_t r, const char *what)
{
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}
inline double AD_Census_batch_rccl(const float *L, const float *R, int pairs, int dispRange, int row, int col, float sigmaC,
                                   float sigmaS, float *dispL, float *dispR, int *devices_used = nullptr, int max_devices = 0)
{
    int G = 0;
    check(smt_device_count(&G), "smt_device_count");
    if (max_devices > 0 && G > max_devices) G = max_devices;
    if (G > pairs) G = pairs;
    if (G <= 0) throw std::runtime_error("no device / empty batch");
    const size_t n = (size_t)row * col;
    const int q = pairs / G, r = pairs % G, cmax = q + (r ? 1 : 0);
    struct Dev {
        smt_adcensus *h = nullptr; void *stream = nullptr;
        float *L = nullptr, *R = nullptr, *send[2] = {nullptr, nullptr}, *recv[2] = {nullptr, nullptr};
        double *sum = nullptr; int start = 0, count = 0;
    };
    std::vector<Dev> dev(G);
    std::vector<int> ids(G);
    for (int g = 0; g < G; g++) ids[g] = g;
    std::vector<ncclComm_t> comm(G);
    rccl_check(ncclCommInitAll(comm.data(), G, ids.data()), "ncclCommInitAll");
    for (int g = 0; g < G; g++) {
        Dev &d = dev[g];
        d.count = q + (g < r ? 1 : 0);
        d.start = g * q + (g < r ? g : r);
        check(smt_set_device(g), "smt_set_device");
        check(smt_adcensus_create_on(g, row, col, dispRange, sigmaC, sigmaS, &d.h), "smt_adcensus_create_on");
        check(smt_stream_create(&d.stream), "smt_stream_create");
        check(smt_adcensus_set_stream(d.h, d.stream), "smt_adcensus_set_stream");
        const size_t bytes = (size_t)d.count * n * sizeof(float), pad = (size_t)cmax * n * sizeof(float);
        check(smt_malloc((void **)&d.L, bytes), "smt_malloc"); check(smt_malloc((void **)&d.R, bytes), "smt_malloc");
        for (int v = 0; v < 2; v++) {
            check(smt_malloc((void **)&d.send[v], pad), "smt_malloc");
            check(smt_malloc((void **)&d.recv[v], pad * G), "smt_malloc");
            check(smt_memset(d.send[v], 0, pad, d.stream), "smt_memset");       // shards are padded to the largest count
        }
        check(smt_malloc((void **)&d.sum, sizeof(double)), "smt_malloc");
        check(smt_memcpy_h2d(d.L, L + (size_t)d.start * n, bytes, d.stream), "h2d");
        check(smt_memcpy_h2d(d.R, R + (size_t)d.start * n, bytes, d.stream), "h2d");
        check(smt_adcensus_compute_batch(d.h, d.L, d.R, d.count, SMT_VIEW_BOTH, d.send[0], d.send[1]), "smt_adcensus_compute_batch");
        check(smt_sum_f32(d.send[0], (size_t)d.count * n, d.sum, d.stream), "smt_sum_f32");
    }
    // the exchange: both views' maps and the checksum, one group
    rccl_check(ncclGroupStart(), "ncclGroupStart");
    for (int g = 0; g < G; g++) {
        Dev &d = dev[g];
        for (int v = 0; v < 2; v++)
            rccl_check(ncclAllGather(d.send[v], d.recv[v], (size_t)cmax * n, ncclFloat, comm[g], (hipStream_t)d.stream), "ncclAllGather");
        rccl_check(ncclAllReduce(d.sum, d.sum, 1, ncclDouble, ncclSum, comm[g], (hipStream_t)d.stream), "ncclAllReduce");
    }
    rccl_check(ncclGroupEnd(), "ncclGroupEnd");
    // device 0 holds everything: trim the padding while copying back
    double checksum = 0.0;
    check(smt_set_device(0), "smt_set_device");
    for (int g = 0; g < G; g++) {
        const size_t bytes = (size_t)dev[g].count * n * sizeof(float);
        check(smt_memcpy_d2h(dispL + (size_t)dev[g].start * n, dev[0].recv[0] + (size_t)g * cmax * n, bytes, dev[0].stream), "d2h");
        check(smt_memcpy_d2h(dispR + (size_t)dev[g].start * n, dev[0].recv[1] + (size_t)g * cmax * n, bytes, dev[0].stream), "d2h");
    }
    check(smt_memcpy_d2h(&checksum, dev[0].sum, sizeof(double), dev[0].stream), "d2h");
    for (int g = 0; g < G; g++) {
        Dev &d = dev[g];
        check(smt_set_device(g), "smt_set_device");
        check(smt_adcensus_status(d.h), "smt_adcensus_status");        // synchronises the device's stream
        smt_free(d.L); smt_free(d.R); smt_free(d.sum);
        for (int v = 0; v < 2; v++) { smt_free(d.send[v]); smt_free(d.recv[v]); }
        smt_adcensus_destroy(d.h);
        smt_stream_destroy(d.stream);
        ncclCommDestroy(comm[g]);
    }
    check(smt_set_device(0), "smt_set_device");
    if (devices_used) *devices_used = G;
    return checksum;
}
#endif  // SMT_HOST_WITH_RCCL

}  // namespace smt
