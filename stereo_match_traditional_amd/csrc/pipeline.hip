// The north-star pipeline of AD-CensusV1/main.cpp:57-92 as one batched entry point ("batch
// variants taking a pair count and strides"; BASELINE.json configs[2] sharded like configs[4]):
//   uchar gray pair -> float copies (:46-55) -> AD_Census both views + WTA (:57-61) -> CrossArmAggregation on
//   the left and on the right image (:67-84) -> ScanlineOptimizer on the LEFT aggregated volume guided by the
//   float left image (:86-89, enabled) -> LeftRightConsistency (:92).
// Nothing here computes: it sequences the library's own entry points and owns the three [H][W][D] volumes
// between the stages, so a batch of pairs reuses them (9.6 GB at 1920x1080x192 whatever the batch size).
// Pairs are independent, which makes this the sharding unit for the pair axis.
// Schedule: everything of a pair runs on the caller's stream except the right view's arms + aggregation, which
// run on an internal stream beside the scanline optimiser of the left view (tools/overlap_probe.py: the
// scanline passes are HBM / latency bound with 2 160 waves in flight, the aggregation is vector-issue bound;
// side by side they take 8.75 ms where one after the other takes 9.65 at 1920x1080 D=192; a high-priority stream
// or raised wave priority for the scanline, or compute units masked out of the aggregation's stream, all measured
// worse or equal).  Event edges per pair:
//   aggregate(left) --ev_left--> side: arms(right), aggregate(right) --ev_right--> main: LR check
// and the next pair's AD-Census (which overwrites the cost volumes) comes after that wait on the main stream.
#include "smt_common.h"
#include <new>

struct smt_pipeline {
    int device;
    int H, W, D;
    smt_pipeline_params P;
    hipStream_t stream;
    hipStream_t side;        // right-view arms + aggregation
    hipEvent_t ev_left, ev_right;
    smt_adcensus *adc;
    smt_crossarm *ca;
    smt_scanline *so;
    float *Lf, *Rf;          // float copies of the current pair
    float *agg[2], *sovol;   // aggregated left / right, scanline sum
};

SMT_API void smt_pipeline_default_params(smt_pipeline_params *p)
{
    if (!p) return;
    p->sigmaC = 10.0f; p->sigmaS = 30.0f;                 // main.cpp:25-26
    p->tao = 30; p->p1 = 10; p->p2 = 150; p->gate = 2;    // main.cpp:27-30
}

SMT_API int smt_pipeline_destroy(smt_pipeline *h)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    if (h->adc) smt_adcensus_destroy(h->adc);
    if (h->ca) smt_crossarm_destroy(h->ca);
    if (h->so) smt_scanline_destroy(h->so);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->ev_left) (void)hipEventDestroy(h->ev_left);
    if (h->ev_right) (void)hipEventDestroy(h->ev_right);
    (void)hipFree(h->Lf); (void)hipFree(h->Rf);
    (void)hipFree(h->agg[0]); (void)hipFree(h->agg[1]); (void)hipFree(h->sovol);
    delete h;
    return SMT_OK;
}

SMT_API int smt_pipeline_create(int H, int W, int D, const smt_pipeline_params *p, smt_pipeline **out)
{
    if (!out || H <= 0 || W <= 0 || D <= 0 || D > 256) return SMT_ERR_ARG;
    smt_pipeline *h = new (std::nothrow) smt_pipeline();
    if (!h) return SMT_ERR_ALLOC;
    h->device = smt_current_device();
    h->H = H; h->W = W; h->D = D;
    if (p) h->P = *p; else smt_pipeline_default_params(&h->P);
    const size_t N = (size_t)H * W, V = N * D;
    smt_crossarm_params cp;
    smt_crossarm_default_params(&cp);
    cp.tau = h->P.tao;
    int rc = smt_adcensus_create(H, W, D, h->P.sigmaC, h->P.sigmaS, &h->adc);
    if (rc == SMT_OK) rc = smt_crossarm_create(H, W, D, &cp, &h->ca);
    if (rc == SMT_OK) rc = smt_scanline_create(H, W, D, h->P.p1, h->P.p2, &h->so);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->Lf, N * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->Rf, N * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->agg[0], V * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->agg[1], V * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->sovol, V * 4);
    if (rc == SMT_OK && hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess) rc = SMT_ERR_HIP;
    if (rc == SMT_OK && hipEventCreateWithFlags(&h->ev_left, hipEventDisableTiming) != hipSuccess) rc = SMT_ERR_HIP;
    if (rc == SMT_OK && hipEventCreateWithFlags(&h->ev_right, hipEventDisableTiming) != hipSuccess) rc = SMT_ERR_HIP;
    if (rc != SMT_OK) { smt_pipeline_destroy(h); return rc; }
    *out = h;
    return SMT_OK;
}

SMT_API int smt_pipeline_create_on(int device, int H, int W, int D, const smt_pipeline_params *p, smt_pipeline **out)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(device);
    return smt_pipeline_create(H, W, D, p, out);
}

SMT_API int smt_pipeline_set_stream(smt_pipeline *h, void *s)
{
    if (!h) return SMT_ERR_ARG;
    h->stream = smt_stream(s);
    int rc = smt_adcensus_set_stream(h->adc, s);
    if (rc == SMT_OK) rc = smt_crossarm_set_stream(h->ca, s);
    if (rc == SMT_OK) rc = smt_scanline_set_stream(h->so, s);
    return rc;
}

SMT_API int smt_pipeline_run_batch(smt_pipeline *h, const uint8_t *grayL, const uint8_t *grayR, int pairs,
                                   float *dispL, float *dispR, uint8_t *cls, int *counts)
{
    if (!h || !grayL || !grayR || pairs <= 0 || !dispL || !dispR || !cls) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    const int H = h->H, W = h->W, D = h->D;
    const size_t N = (size_t)H * W;
    void *st = (void *)h->stream;
    for (int b = 0; b < pairs; b++) {
        const uint8_t *L8 = grayL + b * N, *R8 = grayR + b * N;
        float *dl = dispL + b * N, *dr = dispR + b * N;
        int rc = smt_u8_to_f32(L8, H, W, h->Lf, st);                                    // main.cpp:46-55
        if (rc == SMT_OK) rc = smt_u8_to_f32(R8, H, W, h->Rf, st);
        if (rc == SMT_OK) rc = smt_adcensus_compute(h->adc, h->Lf, h->Rf, SMT_VIEW_BOTH, nullptr, nullptr);   // :57-61 (its WTA maps are overwritten at :75, :84)
        float *vol[2] = {nullptr, nullptr};
        if (rc == SMT_OK) rc = smt_adcensus_volume(h->adc, SMT_VIEW_LEFT, &vol[0]);
        if (rc == SMT_OK) rc = smt_adcensus_volume(h->adc, SMT_VIEW_RIGHT, &vol[1]);
        if (rc == SMT_OK) rc = smt_crossarm_arms(h->ca, L8, 1);                        // :67-72
        if (rc == SMT_OK) rc = smt_crossarm_aggregate(h->ca, vol[0], h->agg[0], 0, nullptr);   // :73 (its WTA :75 is overwritten by :89)
        if (rc == SMT_OK && hipEventRecord(h->ev_left, h->stream) != hipSuccess) rc = SMT_ERR_HIP;
        // right view on the side stream, beside the left view's scanline passes
        if (rc == SMT_OK && hipStreamWaitEvent(h->side, h->ev_left, 0) != hipSuccess) rc = SMT_ERR_HIP;
        if (rc == SMT_OK) rc = smt_crossarm_set_stream(h->ca, (void *)h->side);
        if (rc == SMT_OK) rc = smt_crossarm_arms(h->ca, R8, 1);                        // :77-81 (Initialize again: threshold reset)
        if (rc == SMT_OK) rc = smt_crossarm_aggregate(h->ca, vol[1], h->agg[1], 0, dr);        // :82-84
        if (rc == SMT_OK && hipEventRecord(h->ev_right, h->side) != hipSuccess) rc = SMT_ERR_HIP;
        {
            const int rc2 = smt_crossarm_set_stream(h->ca, st);                        // back, whatever happened
            if (rc == SMT_OK) rc = rc2;
        }
        if (rc == SMT_OK) rc = smt_scanline_run(h->so, h->agg[0], h->Lf, h->sovol, dl);        // :86-89
        if (rc == SMT_OK && hipStreamWaitEvent(h->stream, h->ev_right, 0) != hipSuccess) rc = SMT_ERR_HIP;
        if (rc == SMT_OK) rc = smt_lrcheck(dl, dr, H, W, h->P.gate, cls + b * N, counts ? counts + 2 * b : nullptr, st);   // :92
        if (rc != SMT_OK) { (void)hipStreamSynchronize(h->side); return rc; }
    }
    return SMT_OK;
}

SMT_API int smt_pipeline_volumes(smt_pipeline *h, float **cost_left, float **cost_right, float **agg_left, float **agg_right,
                                 float **scanline_sum)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    int rc = SMT_OK;
    if (cost_left) rc = smt_adcensus_volume(h->adc, SMT_VIEW_LEFT, cost_left);
    if (rc == SMT_OK && cost_right) rc = smt_adcensus_volume(h->adc, SMT_VIEW_RIGHT, cost_right);
    if (agg_left) *agg_left = h->agg[0];
    if (agg_right) *agg_right = h->agg[1];
    if (scanline_sum) *scanline_sum = h->sovol;
    return rc;
}

SMT_API int smt_pipeline_status(smt_pipeline *h)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    const int a = smt_adcensus_status(h->adc), c = smt_crossarm_status(h->ca);
    return a != SMT_OK ? a : c;
}
