// The north-star pipeline of AD-CensusV1/main.cpp:57-92 as one batched entry point ("batch
// variants taking a pair count and strides"; BASELINE.json configs[2] sharded like configs[4]):
//   uchar gray pair -> float copies (:46-55) -> AD_Census both views + WTA (:57-61) -> CrossArmAggregation on
//   the left and on the right image (:67-84) -> ScanlineOptimizer on the LEFT aggregated volume guided by the
//   float left image (:86-89, enabled) -> LeftRightConsistency (:92).
// Nothing here computes: it sequences the library's own entry points and owns the three [H][W][D] volumes
// between the stages, so a batch of pairs reuses them (9.6 GB at 1920x1080x192 whatever the batch size).
// Pairs are independent, which makes this the sharding unit for the pair axis.
// Schedules.  SMT_PIPE_SCHEDULE=2: three streams and double-buffered front-end state, so that kernels with
// different bottlenecks run side by side (tools/overlap_probe.py: the scanline passes are HBM / latency bound with
// 2 160 waves in flight, the aggregation is vector-issue bound; side by side they take 8.75 ms where one after the
// other takes 9.65 at 1920x1080 D=192; a high-priority stream or raised wave priority for the scanline, or compute
// units masked out of the aggregation's stream, all measured worse or equal):
//   front  F : u8 -> f32, AD-Census of pair b into table / volume / float-image set b & 1   (store bound)
//   main   M : arms + aggregation of the left view, LR check of pair b - 1, scanline of pair b
//   side   S : arms + aggregation of the right view (its own crossarm handle), beside the scanline of the same pair
// Event edges for pair b, set s = b & 1:
//   M: in      --> F                       the caller's inputs are ordered before everything
//   M: scan(b-2) done, S: right(b-2) done --> F     set s is free again (float images, cost volumes)
//   F: front(b) --> M                      arms(left), aggregate(left) --ev_left--> S: arms(right), aggregate(right)
//   S: right(b-1) --> M                    LR check of pair b - 1, issued after aggregate(left, b): the right view of
//                                          b - 1 has had that whole aggregation to finish, M never waits for it
//   after the loop: S: right(last) --> M: LR check(last).  Every call leaves all its work ordered on the caller's stream.
// SMT_PIPE_SCHEDULE=1 (the default) is the two-stream form: both views' arms beside the AD-Census of the same pair and
// the right view's aggregation (its own crossarm handle) beside the left view's scanline, nothing else double-buffered; SMT_PIPE_SCHEDULE=0 runs everything on the caller's stream.
// Measured at 1920x1080 D=192, 8 pairs per call, ms per pair: 10.7 (0), 9.66-9.76 (1), 9.61-9.66 (2) -- the third
// stream buys 1 % for a second AD-Census handle (3.3 GB) and a second crossarm handle, hence the default.
#include "smt_common.h"
#include <new>
#include <stdlib.h>

struct smt_pipeline {
    int device;
    int H, W, D;
    smt_pipeline_params P;
    int sched;               // 0, 1, 2: see above
    int last_set;            // set of the last pair run (smt_pipeline_volumes)
    hipStream_t stream;      // M: the caller's stream
    hipStream_t side, front; // S, F
    hipEvent_t ev_in, ev_left, ev_front[2], ev_scan[2], ev_right[2];
    smt_adcensus *adc[2];
    smt_crossarm *caL, *caR;
    smt_scanline *so;
    float *Lf[2], *Rf[2];    // float copies of the pair in flight, per set
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
    (void)hipDeviceSynchronize();
    for (int k = 0; k < 2; k++) {
        if (h->adc[k]) smt_adcensus_destroy(h->adc[k]);
        if (h->ev_front[k]) (void)hipEventDestroy(h->ev_front[k]);
        if (h->ev_scan[k]) (void)hipEventDestroy(h->ev_scan[k]);
        if (h->ev_right[k]) (void)hipEventDestroy(h->ev_right[k]);
        (void)hipFree(h->Lf[k]); (void)hipFree(h->Rf[k]);
        (void)hipFree(h->agg[k]);
    }
    if (h->caL) smt_crossarm_destroy(h->caL);
    if (h->caR) smt_crossarm_destroy(h->caR);
    if (h->so) smt_scanline_destroy(h->so);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    if (h->ev_left) (void)hipEventDestroy(h->ev_left);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->front) (void)hipStreamDestroy(h->front);
    (void)hipFree(h->sovol);
    delete h;
    return SMT_OK;
}

static int pipeline_apply_streams(smt_pipeline *h)
{
    void *m = (void *)h->stream;
    int rc = smt_crossarm_set_stream(h->caL, m);
    if (rc == SMT_OK) rc = smt_scanline_set_stream(h->so, m);
    if (rc == SMT_OK && h->caR) rc = smt_crossarm_set_stream(h->caR, (void *)h->side);
    for (int k = 0; k < 2 && rc == SMT_OK; k++)
        if (h->adc[k]) rc = smt_adcensus_set_stream(h->adc[k], h->sched == 2 ? (void *)h->front : m);
    return rc;
}

SMT_API int smt_pipeline_create(int H, int W, int D, const smt_pipeline_params *p, smt_pipeline **out)
{
    if (!out || H <= 0 || W <= 0 || D <= 0 || D > SMT_MAX_DISPARITY) return SMT_ERR_ARG;
    smt_pipeline *h = new (std::nothrow) smt_pipeline();
    if (!h) return SMT_ERR_ALLOC;
    h->device = smt_current_device();
    h->H = H; h->W = W; h->D = D;
    if (p) h->P = *p; else smt_pipeline_default_params(&h->P);
    {
        const char *e = getenv("SMT_PIPE_SCHEDULE");
        h->sched = (e && e[0] >= '0' && e[0] <= '2' && !e[1]) ? e[0] - '0' : 1;
    }
    const int nset = h->sched == 2 ? 2 : 1;
    const size_t N = (size_t)H * W, V = N * D;
    smt_crossarm_params cp;
    smt_crossarm_default_params(&cp);
    cp.tau = h->P.tao;
    int rc = SMT_OK;
    for (int k = 0; k < nset && rc == SMT_OK; k++) {
        rc = smt_adcensus_create(H, W, D, h->P.sigmaC, h->P.sigmaS, &h->adc[k]);
        if (rc == SMT_OK) rc = smt_malloc((void **)&h->Lf[k], N * 4);
        if (rc == SMT_OK) rc = smt_malloc((void **)&h->Rf[k], N * 4);
    }
    if (rc == SMT_OK) rc = smt_crossarm_create(H, W, D, &cp, &h->caL);
    if (rc == SMT_OK && h->sched >= 1) rc = smt_crossarm_create(H, W, D, &cp, &h->caR);
    if (rc == SMT_OK) rc = smt_scanline_create(H, W, D, h->P.p1, h->P.p2, &h->so);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->agg[0], V * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->agg[1], V * 4);
    if (rc == SMT_OK) rc = smt_malloc((void **)&h->sovol, V * 4);
    if (rc == SMT_OK && hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess) rc = SMT_ERR_HIP;
    if (rc == SMT_OK && hipStreamCreateWithFlags(&h->front, hipStreamNonBlocking) != hipSuccess) rc = SMT_ERR_HIP;
    hipEvent_t *evs[] = {&h->ev_in, &h->ev_left, &h->ev_front[0], &h->ev_front[1], &h->ev_scan[0], &h->ev_scan[1],
                         &h->ev_right[0], &h->ev_right[1]};
    for (hipEvent_t *e : evs)
        if (rc == SMT_OK && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) rc = SMT_ERR_HIP;
    if (rc == SMT_OK) rc = pipeline_apply_streams(h);
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
    smt_dev_guard dev_guard(h->device);
    h->stream = smt_stream(s);
    return pipeline_apply_streams(h);
}

#define PIPE_HIP(call) do { if (rc == SMT_OK && (call) != hipSuccess) rc = SMT_ERR_HIP; } while (0)

// schedules 0 and 1: one set of front-end state, the left view's handle does both views
static int pipeline_run_simple(smt_pipeline *h, const uint8_t *grayL, const uint8_t *grayR, int pairs, float *dispL,
                               float *dispR, uint8_t *cls, int *counts)
{
    const int H = h->H, W = h->W;
    const size_t N = (size_t)H * W;
    void *st = (void *)h->stream;
    const bool two = h->sched == 1;
    if (two) {
        int rc = SMT_OK;
        PIPE_HIP(hipEventRecord(h->ev_in, h->stream));
        PIPE_HIP(hipStreamWaitEvent(h->side, h->ev_in, 0));                            // the caller's inputs, for the side stream
        if (rc != SMT_OK) return rc;
    }
    for (int b = 0; b < pairs; b++) {
        const uint8_t *L8 = grayL + b * N, *R8 = grayR + b * N;
        float *dl = dispL + b * N, *dr = dispR + b * N;
        int rc = SMT_OK;
        if (two) {
            // both views' arms need only the images: on the side stream (the right view has its own handle), beside
            // this pair's AD-Census
            rc = smt_crossarm_set_stream(h->caL, (void *)h->side);
            if (rc == SMT_OK) rc = smt_crossarm_arms(h->caL, L8, 1);                   // :67-72
            PIPE_HIP(hipEventRecord(h->ev_front[0], h->side));
            const int rc2 = smt_crossarm_set_stream(h->caL, st);
            if (rc == SMT_OK) rc = rc2;
            if (rc == SMT_OK) rc = smt_crossarm_arms(h->caR, R8, 1);                   // :77-81 (its own Initialize: threshold reset)
        }
        if (rc == SMT_OK) rc = smt_u8_to_f32(L8, H, W, h->Lf[0], st);                   // main.cpp:46-55
        if (rc == SMT_OK) rc = smt_u8_to_f32(R8, H, W, h->Rf[0], st);
        if (rc == SMT_OK) rc = smt_adcensus_compute(h->adc[0], h->Lf[0], h->Rf[0], SMT_VIEW_BOTH, nullptr, nullptr);   // :57-61 (its WTA maps are overwritten at :75, :84)
        float *vol[2] = {nullptr, nullptr};
        if (rc == SMT_OK) rc = smt_adcensus_volume(h->adc[0], SMT_VIEW_LEFT, &vol[0]);
        if (rc == SMT_OK) rc = smt_adcensus_volume(h->adc[0], SMT_VIEW_RIGHT, &vol[1]);
        if (two) PIPE_HIP(hipStreamWaitEvent(h->stream, h->ev_front[0], 0));
        else if (rc == SMT_OK) rc = smt_crossarm_arms(h->caL, L8, 1);                  // :67-72
        if (rc == SMT_OK) rc = smt_crossarm_aggregate(h->caL, vol[0], h->agg[0], 0, nullptr);  // :73 (its WTA :75 is overwritten by :89)
        if (two) {
            PIPE_HIP(hipEventRecord(h->ev_left, h->stream));
            PIPE_HIP(hipStreamWaitEvent(h->side, h->ev_left, 0));
            if (rc == SMT_OK) rc = smt_crossarm_aggregate(h->caR, vol[1], h->agg[1], 0, dr);   // :82-84, beside the scanline
            PIPE_HIP(hipEventRecord(h->ev_right[0], h->side));
        } else {
            if (rc == SMT_OK) rc = smt_crossarm_arms(h->caL, R8, 1);                   // :77-81 (Initialize again: threshold reset)
            if (rc == SMT_OK) rc = smt_crossarm_aggregate(h->caL, vol[1], h->agg[1], 0, dr);   // :82-84
        }
        if (rc == SMT_OK) rc = smt_scanline_run(h->so, h->agg[0], h->Lf[0], h->sovol, dl);     // :86-89
        if (two) PIPE_HIP(hipStreamWaitEvent(h->stream, h->ev_right[0], 0));
        if (rc == SMT_OK) rc = smt_lrcheck(dl, dr, H, W, h->P.gate, cls + b * N, counts ? counts + 2 * b : nullptr, st);   // :92
        if (rc != SMT_OK) { (void)hipStreamSynchronize(h->side); return rc; }
    }
    h->last_set = 0;
    return SMT_OK;
}

SMT_API int smt_pipeline_run_batch(smt_pipeline *h, const uint8_t *grayL, const uint8_t *grayR, int pairs,
                                   float *dispL, float *dispR, uint8_t *cls, int *counts)
{
    if (!h || !grayL || !grayR || pairs <= 0 || !dispL || !dispR || !cls) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    if (h->sched != 2) return pipeline_run_simple(h, grayL, grayR, pairs, dispL, dispR, cls, counts);
    const int H = h->H, W = h->W;
    const size_t N = (size_t)H * W;
    hipStream_t M = h->stream, S = h->side, F = h->front;
    int rc = SMT_OK;
    PIPE_HIP(hipEventRecord(h->ev_in, M));
    PIPE_HIP(hipStreamWaitEvent(F, h->ev_in, 0));
    for (int b = 0; b < pairs && rc == SMT_OK; b++) {
        const int s = b & 1;
        const uint8_t *L8 = grayL + b * N, *R8 = grayR + b * N;
        float *dl = dispL + b * N, *dr = dispR + b * N;
        // F: front end of pair b into set s
        if (b >= 2) {
            PIPE_HIP(hipStreamWaitEvent(F, h->ev_scan[s], 0));     // scanline(b-2) read Lf[s]; aggregate(left, b-2) read the volumes
            PIPE_HIP(hipStreamWaitEvent(F, h->ev_right[s], 0));    // aggregate(right, b-2) read the volumes
        }
        if (rc == SMT_OK) rc = smt_u8_to_f32(L8, H, W, h->Lf[s], (void *)F);                   // main.cpp:46-55
        if (rc == SMT_OK) rc = smt_u8_to_f32(R8, H, W, h->Rf[s], (void *)F);
        if (rc == SMT_OK) rc = smt_adcensus_compute(h->adc[s], h->Lf[s], h->Rf[s], SMT_VIEW_BOTH, nullptr, nullptr);   // :57-61
        PIPE_HIP(hipEventRecord(h->ev_front[s], F));
        float *vol[2] = {nullptr, nullptr};
        if (rc == SMT_OK) rc = smt_adcensus_volume(h->adc[s], SMT_VIEW_LEFT, &vol[0]);
        if (rc == SMT_OK) rc = smt_adcensus_volume(h->adc[s], SMT_VIEW_RIGHT, &vol[1]);
        // M: left view
        PIPE_HIP(hipStreamWaitEvent(M, h->ev_front[s], 0));
        if (rc == SMT_OK) rc = smt_crossarm_arms(h->caL, L8, 1);                               // :67-72
        if (rc == SMT_OK) rc = smt_crossarm_aggregate(h->caL, vol[0], h->agg[0], 0, nullptr);  // :73 (its WTA :75 is overwritten by :89)
        PIPE_HIP(hipEventRecord(h->ev_left, M));
        if (b >= 1) {                                                                           // :92 of pair b - 1
            PIPE_HIP(hipStreamWaitEvent(M, h->ev_right[s ^ 1], 0));
            if (rc == SMT_OK)
                rc = smt_lrcheck(dl - N, dr - N, H, W, h->P.gate, cls + (b - 1) * N, counts ? counts + 2 * (b - 1) : nullptr, (void *)M);
        }
        if (rc == SMT_OK) rc = smt_scanline_run(h->so, h->agg[0], h->Lf[s], h->sovol, dl);     // :86-89
        PIPE_HIP(hipEventRecord(h->ev_scan[s], M));
        // S: right view, beside the scanline passes
        PIPE_HIP(hipStreamWaitEvent(S, h->ev_left, 0));
        if (rc == SMT_OK) rc = smt_crossarm_arms(h->caR, R8, 1);                               // :77-81
        if (rc == SMT_OK) rc = smt_crossarm_aggregate(h->caR, vol[1], h->agg[1], 0, dr);       // :82-84
        PIPE_HIP(hipEventRecord(h->ev_right[s], S));
    }
    if (rc == SMT_OK) {
        const int b = pairs - 1, s = b & 1;
        PIPE_HIP(hipStreamWaitEvent(M, h->ev_right[s], 0));
        if (rc == SMT_OK)
            rc = smt_lrcheck(dispL + b * N, dispR + b * N, H, W, h->P.gate, cls + b * N, counts ? counts + 2 * b : nullptr, (void *)M);
        h->last_set = s;
    }
    if (rc != SMT_OK) { (void)hipStreamSynchronize(S); (void)hipStreamSynchronize(F); }
    return rc;
}

SMT_API int smt_pipeline_volumes(smt_pipeline *h, float **cost_left, float **cost_right, float **agg_left, float **agg_right,
                                 float **scanline_sum)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    int rc = SMT_OK;
    smt_adcensus *a = h->adc[h->last_set];
    if (cost_left) rc = smt_adcensus_volume(a, SMT_VIEW_LEFT, cost_left);
    if (rc == SMT_OK && cost_right) rc = smt_adcensus_volume(a, SMT_VIEW_RIGHT, cost_right);
    if (agg_left) *agg_left = h->agg[0];
    if (agg_right) *agg_right = h->agg[1];
    if (scanline_sum) *scanline_sum = h->sovol;
    return rc;
}

SMT_API int smt_pipeline_status(smt_pipeline *h)
{
    if (!h) return SMT_ERR_ARG;
    smt_dev_guard dev_guard(h->device);
    (void)hipStreamSynchronize(h->stream);               // everything of the last call is ordered on the caller's stream
    int rc = SMT_OK;
    for (int k = 0; k < 2; k++)
        if (h->adc[k]) { const int a = smt_adcensus_status(h->adc[k]); if (rc == SMT_OK) rc = a; }
    { const int c = smt_crossarm_status(h->caL); if (rc == SMT_OK) rc = c; }
    if (h->caR) { const int c = smt_crossarm_status(h->caR); if (rc == SMT_OK) rc = c; }
    return rc;
}
