/*
 * smt.h -- C ABI of the MI355X-native dense stereo cost-volume engine (libsmt_hip.so).
 *
 * This is the drop-in boundary for the per-pixel x per-disparity hot path of
 * Asherchi/Stereo_Match_Traditional.  The reference has no FFI of its own: its
 * boundary is the set of C++ entry points its five main()s call with raw buffers
 * (SURVEY.md 8b).  Each entry point below names the reference function(s) it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the C++ shim a
 * maintainer of the reference would add to route those calls here.
 *
 * Conventions
 *   - plain C types only; every image / map / volume pointer is a DEVICE pointer (HIP
 *     global memory) owned by the caller unless stated otherwise.  smt_malloc /
 *     smt_memcpy_* are provided for hosts that have no allocator of their own.
 *   - images  [H][W] row-major; volumes [H][W][D], d fastest, float32 -- the reference
 *     layout (AD-CensusV1/AD-Census.h:87).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     asynchronous on that stream unless documented as synchronising.
 *   - every function returns SMT_OK (0) or a negative smt_status.  The reference
 *     validates nothing (void functions, UB on bad sizes); this ABI rejects bad
 *     arguments instead.
 *   - handles are thread-compatible, not thread-safe (the reference objects hold
 *     mutable state too, e.g. CrossArm.h:34 `_tao`).
 *   - reference defects that change results are reproduced by default; see the
 *     SMT_QUIRK_* flags.
 */
#ifndef SMT_H_
#define SMT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMT_VERSION 100 /* 0.1.0 */

typedef enum smt_status {
    SMT_OK = 0,
    SMT_ERR_ARG = -1,     /* null pointer / non-positive size / unsupported parameter */
    SMT_ERR_HIP = -2,     /* a HIP runtime call failed (smt_last_hip_error has the code) */
    SMT_ERR_ALLOC = -3,   /* device allocation failed */
    SMT_ERR_DOMAIN = -4,  /* image values outside the integer 0..255 domain the path assumes */
    SMT_ERR_REF_UB = -5,  /* inputs for which the reference's behaviour is undefined */
    SMT_ERR_STATE = -6    /* call order violated (e.g. aggregate before arms) */
} smt_status;

/* Reference-defect switches.  Default (0) = reference-faithful. */
#define SMT_QUIRK_FIX_RIGHT_ARM_STRIDE 0x1u /* undo `col = _row` in ComputeRightArmLength (CrossArm.cpp:265) */

/* views bit mask */
#define SMT_VIEW_LEFT 1
#define SMT_VIEW_RIGHT 2
#define SMT_VIEW_BOTH 3

const char *smt_strerror(int status);
int smt_version(void);
int smt_last_hip_error(void);
int smt_device_count(int *count);
int smt_set_device(int device);
/* Devices.  A handle lives on the HIP device that was current in the calling thread when it was created
 * (smt_*_create) or on the one named explicitly (smt_*_create_on(device, ...), same arguments otherwise);
 * its entry points make that device current for the duration of the call and restore the caller's, so
 * one host thread can drive a handle per GPU.  Buffers and streams passed to a handle must belong to its
 * device.  The stateless entry points (smt_wta, smt_lrcheck, smt_sad, ...) run on the device that is
 * current in the calling thread.
 *
 * Limits (the reference has none; outside them SMT_ERR_ARG):  dispRange D <= SMT_MAX_DISPARITY = 512 on the
 * AD-Census pipeline (smt_adcensus_*, smt_wta, smt_crossarm_*, smt_scanline_*, smt_pipeline_*: one wavefront
 * spans the disparity axis, a lane owns up to 8 consecutive hypotheses; the tuned kernels cover D <= 256, beyond
 * it the first-version kernels run -- the table-lookup cost kernel, the one-pixel-per-wave rectangle walk, the
 * predicated scanline passes -- with the same results, untuned) and D <= 256 on the window matchers, the
 * CrossAggregator and the CBLSM helpers (smt_sad, smt_ncc, smt_asw, smt_crossagg_*, smt_cblsm_*); ASW window side
 * 2*(winSize+1)+1 <= 64 (winSize <= 30; one window row per wavefront pass); MedianFilter wnd_size <= 7;
 * volumes of 4 GiB or more take the plain one-pixel-per-wave aggregation kernel (32-bit tap offsets in
 * the shared-tap kernels). */

#define SMT_MAX_DISPARITY 512

/* ---- device memory helpers (plumbing; not part of the reference's surface) ---------- */
int smt_malloc(void **dptr, size_t bytes);
int smt_free(void *dptr);
int smt_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int smt_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int smt_memset(void *dst_dev, int byte, size_t bytes, void *stream);
int smt_stream_create(void **stream);
int smt_stream_destroy(void *stream);
int smt_stream_sync(void *stream); /* synchronising */

/* float64 sum of a float32 device array into *out_dev (device double; zeroed on the stream first) -- the
 * per-device term of the gather checksum of the batched multi-GPU configuration. */
int smt_sum_f32(const float *x, size_t n, double *out_dev, void *stream);

/* ---- kernel timing (bench.py roofline leg) ----------------------------------------------
 * When enabled, every N-th pair processed on the handle records HIP events -- around the table
 * kernels and around the cost kernel(s), each on the stream the kernels are launched on -- into a ring
 * of `SMT_TIMING_SLOTS` slots (a pair whose tables were built inside the previous pair's cost launch,
 * smt_adcensus_compute_batch, reports ~0 for them); nothing synchronises until
 * smt_adcensus_kernel_times, which waits for the last event and returns the per-pair
 * durations in milliseconds, oldest first. */
#define SMT_TIMING_SLOTS 1024
typedef struct smt_adcensus smt_adcensus;
int smt_adcensus_timing(smt_adcensus *h, int enable); /* 0 off, N > 0: record every N-th pair; also clears the ring */
int smt_adcensus_kernel_times(smt_adcensus *h, float *prep_ms, float *cost_ms, int capacity,
                              int *count);

/* Measurement hook (bench.py roofline leg; not part of the reference's surface).  Needs a pair already
 * computed on the handle and D = 64, 128, 192 or 256.  On the handle's stream it runs `reps` launches of
 * a store-only twin of the both-views cost kernel (identical grid, chunk order and streaming stores into
 * the handle's own volumes, no arithmetic) and then `reps` launches of the real kernel with in-kernel
 * counter stamps, and returns
 *   sclk_mhz       median over workgroups of d(s_memtime)/d(s_memrealtime) x 100 MHz inside the last
 *                  stamped launch = the shader clock the cost kernel actually ran at,
 *   cost_ms        mean duration of the stamped launches (no WTA maps are written),
 *   store_only_ms  mean duration of the store-only launches = the store ceiling of this pattern in
 *                  this process on these buffers.
 * Any of the three may be NULL.  The volumes hold the last pair's costs again on return.  Synchronising. */
int smt_adcensus_diag(smt_adcensus *h, int reps, float *sclk_mhz, float *cost_ms, float *store_only_ms);

/* =====================================================================================
 * AD-Census cost volume + WTA        replaces class AD_Census, AD-CensusV1/AD-Census.h
 * ===================================================================================== */

/* AD_Census::Initialize (AD-Census.h:322-344): fixes H, W, D, sigmaC (AD, `_sigmaC`) and
 * sigmaS (census, `_sigmaS`); allocates the left and right cost volumes (costVolume,
 * costVolumeRight) plus census tables.  The reference's separate AD / census volumes
 * (ADcostVolum, CensusVolum, ...) are never materialised. */
int smt_adcensus_create(int H, int W, int D, float sigmaC, float sigmaS, smt_adcensus **out);
int smt_adcensus_create_on(int device, int H, int W, int D, float sigmaC, float sigmaS, smt_adcensus **out);
/* The same with the two measuring steps of Initialize under the caller's control (device < 0: the current one).
 * Without flags smt_adcensus_create allocates up to six candidate pairs of volumes (transiently 6 x 2 x 4*H*W*D
 * bytes: 19 GB at 1920x1080x192) to pick the placement with the fastest stores, and times the cost kernel with
 * streaming and with ordinary stores -- tens of milliseconds, worth it for a handle that lives for a batch, wrong
 * for a caller that creates a handle per request or shares the device:
 *   SMT_ADCENSUS_NO_PLACEMENT_SEARCH   keep the first allocation of the volumes
 *   SMT_ADCENSUS_NO_STORE_CALIBRATION  streaming stores without timing the alternative
 * (SMT_PLACEMENT=0 / SMT_STORE_MODE in the environment still do the same process-wide.) */
#define SMT_ADCENSUS_NO_PLACEMENT_SEARCH 0x1u
#define SMT_ADCENSUS_NO_STORE_CALIBRATION 0x2u
int smt_adcensus_create_ex(int device, int H, int W, int D, float sigmaC, float sigmaS, unsigned flags, smt_adcensus **out);
int smt_adcensus_destroy(smt_adcensus *h);
/* How smt_adcensus_create placed the two volumes: it allocates up to 6 candidate pairs, times a
 * store-only twin of the cost kernel on each and keeps the fastest (the HBM write rate of the same
 * kernel differs by ~18 % between allocations); SMT_PLACEMENT=0 in the
 * environment disables the search.  tries = candidate pairs allocated, store_only_ms = the kept pair's
 * store-only time (0 when there was no search).  Either pointer may be NULL. */
int smt_adcensus_placement(smt_adcensus *h, int *tries, float *store_only_ms);
/* Which stores the both-views cost kernel uses on this handle: smt_adcensus_create times the kernel a few
 * launches with streaming (non-temporal) and with ordinary stores and keeps ordinary ones only when they win by
 * more than 2 % (a device property like the placement; SMT_STORE_MODE=nt|plain in the environment fixes it).
 * plain = 1 / 0, nt_ms / plain_ms = the calibration times (0 when there was none).  Any pointer may be NULL. */
int smt_adcensus_store_mode(smt_adcensus *h, int *plain, float *nt_ms, float *plain_ms);
int smt_adcensus_set_stream(smt_adcensus *h, void *stream);

/* ComputeADcensus (AD-Census.h:271-294) for SMT_VIEW_LEFT, ComputeADcensusRight
 * (:296-318) for SMT_VIEW_RIGHT, followed -- when dispL / dispR are non-NULL -- by
 * AD_Census::WTA (:346-380) fused into the same kernel.
 *   L, R        float32 [H][W], integer-valued 0..255 (main.cpp:46-55 builds them from
 *               uchar gray images); anything else raises SMT_ERR_DOMAIN at the next
 *               smt_adcensus_status.
 *   dispL/R     float32 [H][W] out, integer-valued, may be NULL. */
int smt_adcensus_compute(smt_adcensus *h, const float *L, const float *R, int views,
                         float *dispL, float *dispR);

/* Same, for a batch of `pairs` image pairs laid out [pairs][H][W]; the volumes are
 * reused per pair (only the last pair's stay readable), the disparity maps are
 * [pairs][H][W].  This is the sharding unit of the multi-GPU configuration.  The census tables
 * are double-buffered inside the handle: with both views and D <= 256 the table workgroups of pair
 * b+1 are spread through the grid of pair b's cost launch (one launch per pair, one stream); single
 * views and D > 256 build them on an internal stream beside pair b's cost kernel.  SMT_OVERLAP =
 * 0 / 1 / 2 in the environment forces in-order / internal-stream / fused (read at every call). */
int smt_adcensus_compute_batch(smt_adcensus *h, const float *L, const float *R, int pairs,
                               int views, float *dispL, float *dispR);

/* GetPtrLeft / GetPtrRight (AD-Census.h:50-72): borrowed device pointer, valid until
 * destroy. view = SMT_VIEW_LEFT or SMT_VIEW_RIGHT. */
int smt_adcensus_volume(smt_adcensus *h, int view, float **vol);

/* Test hook: on != 0 routes the pair through the first-version table-lookup kernel (a second,
 * independent formulation kept for cross-checking) instead of the register-window kernel. */
int smt_adcensus_force_generic(smt_adcensus *h, int on);

/* Test hook, host only (no GPU): checks the workgroup arithmetic of the fused batch launch -- with `ncost` cost
 * workgroups (a multiple of 8) and `nprep` table workgroups every cost group and every table group is reached exactly
 * once.  SMT_OK, or SMT_ERR_STATE if the mapping is not a bijection. */
int smt_adcensus_selftest_fused_grid(int ncost, int nprep);

/* Synchronises the stream and returns SMT_ERR_DOMAIN if any pixel seen since the previous
 * smt_adcensus_status call (or since create) was not an integer in 0..255 (then those pairs' volumes
 * are unspecified), else SMT_OK.  Read-and-clear: a bad pair does not poison later checks.  Bad input is
 * therefore reported late, at the first status call after the compute; call it before consuming results
 * (the host mirrors do so in WTA() / GetPtr*()). */
int smt_adcensus_status(smt_adcensus *h);

/* First-strict-minimum argmin over d of one volume.  Replaces
 * CrossArmAggregation::WTA (CrossArm.cpp:33-57), ScanlineOptimizer::WTA
 * (ScanlineOptimizer.h:40-64), ComputeDispOringin (CBLSM/CBLSM.h:383-407) and, applied
 * twice, AD_Census::WTA. */
int smt_wta(const float *vol, int H, int W, int D, float *disp, void *stream);

/* =====================================================================================
 * Cross-arm rectangle aggregation     replaces class CrossArmAggregation
 *                                     (AD-CensusV1/CrossArm.{h,cpp}) and the active
 *                                     CBLSM.h functions ArmLength{L,R,Up,Down},
 *                                     costAggregationV5
 * ===================================================================================== */
typedef struct smt_crossarm smt_crossarm;

typedef struct smt_crossarm_params {
    int tau;          /* initial threshold: 30 (main.cpp:27) / 25 (CBLSM.cpp:30) */
    int tau_low;      /* 6  (CrossArm.cpp:225, CBLSM.h:719) */
    int sec_length;   /* 17 (CrossArm.cpp:223) / secLength (CBLSM.cpp:32) */
    int max_length;   /* 34 (CrossArm.cpp:226) / maxLength (CBLSM.cpp:31) */
    int chain_tau;    /* 1: `_tao` is a member, sticky across the four direction calls
                            (CrossArm.h:34); 0: by-value per call (CBLSM.h:643) */
    unsigned quirks;  /* SMT_QUIRK_*; 0 = faithful.  CBLSM-style arms have no stride bug:
                            pass SMT_QUIRK_FIX_RIGHT_ARM_STRIDE for them. */
} smt_crossarm_params;

void smt_crossarm_default_params(smt_crossarm_params *p); /* AD-CensusV1 main.cpp values */
void smt_crossarm_cblsm_params(smt_crossarm_params *p);   /* CBLSM.cpp values */

/* CrossArmAggregation::Initialize (CrossArm.cpp:6-18): allocates the four arm maps and
 * resets the sticky threshold. */
int smt_crossarm_create(int H, int W, int D, const smt_crossarm_params *p, smt_crossarm **out);
int smt_crossarm_create_on(int device, int H, int W, int D, const smt_crossarm_params *p, smt_crossarm **out);
int smt_crossarm_destroy(smt_crossarm *h);
int smt_crossarm_set_stream(smt_crossarm *h, void *stream);

/* ComputeLeftArmLength, ComputeRightArmLength, ComputeTopArmLength,
 * ComputeButtonArmLength (CrossArm.cpp:147-598) in that order with the threshold state
 * chained as the reference's member does.  Resets the threshold first, i.e. it is
 * Initialize + the four calls of main.cpp:68-72.
 *   img        uint8 [H][W][channels], channels 1 (gray branch) or 3 (Vec3b branch). */
int smt_crossarm_arms(smt_crossarm *h, const uint8_t *img, int channels);

/* One-to-one forms of the reference's four calls (CrossArm.h:15-18), for callers that run a subset or
 * another order -- which changes the sticky-threshold chain (`_tao` is lowered at CrossArm.cpp:223-225 by
 * whichever call first walks past sec_length and stays lowered for every later pixel and call):
 *   smt_crossarm_reset    = the state part of Initialize (CrossArm.cpp:13-17): threshold back to tau, the
 *                           four maps zeroed;
 *   smt_crossarm_arm_dir  = ComputeLeftArmLength (dir 0, :147-260), ComputeRightArmLength (1, :262-373),
 *                           ComputeTopArmLength (2, :375-486), ComputeButtonArmLength (3, :488-598) with
 *                           the threshold as the previous call left it;
 *   smt_crossarm_tau      = the current `_tao` (synchronising; for tests).
 * smt_crossarm_arms(h, img, ch) == reset + arm_dir 0, 1, 2, 3, in two launches.  With chain_tau = 0
 * (CBLSM.h:643, by-value threshold) every call starts from tau. */
int smt_crossarm_reset(smt_crossarm *h);
int smt_crossarm_arm_dir(smt_crossarm *h, const uint8_t *img, int channels, int dir);
int smt_crossarm_tau(smt_crossarm *h, int *tau);

/* Test hook: on != 0 computes arms with the first-version kernels (neighbour-by-neighbour walk, an
 * independent formulation) instead of the bit-mask kernels; they are also what runs when sec_length or
 * max_length exceeds 63. */
int smt_crossarm_set_arm_walk(smt_crossarm *h, int on);

/* Borrowed pointers to the int32 [H][W] arm maps (leftLength, rightLength, topLength,
 * buttonLenght; CrossArm.h:30-33). */
int smt_crossarm_arm_maps(smt_crossarm *h, int **left, int **right, int **top, int **bottom);
/* The other direction: arm maps the caller already has (DEVICE int32 [H][W] each) become the handle's maps, so that
 * the aggregation entry below serves callers whose reference signature takes the four arrays --
 * costAggregationV5(dispvolume, CostVolume, ArmvolumeL, ArmvolumeR, ArmvolumeUp, ArmvolumeDown, ...) (CBLSM.h:1179);
 * CBLSM.cpp:150 aggregates the RIGHT view's volume with the LEFT image's arms this way.  Lengths outside 0..8191
 * are clamped and make smt_crossarm_status return SMT_ERR_REF_UB. */
int smt_crossarm_load_arm_maps(smt_crossarm *h, const int *left, const int *right, const int *top, const int *bottom);

/* order 0: AggregationVertical (CrossArm.cpp:60-102), columns outer / rows inner;
 * order 1: costAggregationV5 (CBLSM.h:1179-1224), rows outer / columns inner;
 * order 2: Aggregation (CrossArm.cpp:104-145; public in CrossArm.h:19, no call site): rows outer with
 *          EXCLUSIVE upper bounds [-up, down) x [-L, R); a pixel whose rectangle is empty divides 0 by 0
 *          (:138) -- the result is NaN there and smt_crossarm_status returns SMT_ERR_REF_UB.
 * Sequential float adds in exactly that order, divided by the tap count.
 * If disp != NULL the WTA of the aggregated volume is fused (CrossArm.cpp:33-57).
 * Returns SMT_ERR_REF_UB from smt_crossarm_status when a rectangle leaves the plane
 * (possible with the right-arm stride bug on small / non-landscape images). */
int smt_crossarm_aggregate(smt_crossarm *h, const float *vol_in, float *vol_out, int order,
                           float *disp);
int smt_crossarm_status(smt_crossarm *h); /* synchronising; read-and-clear: reports rectangles that left the
                                             plane (or, order 2, were empty) since the previous status call */
/* Test / tuning hook: which aggregation kernel runs.  13 (default) = 12 with the scalar side of a tap (byte offsets of
 * its two flag rows, one "group has a member" bit per tile row) packed into one word by the vector classification of
 * the batch; 12 = 4x4 pixels per wave, every tap of the union of
 * their rectangles loaded once and added under membership flags (v_pk_fma_f32, flag pairs in SGPRs), tile rows
 * without a member skipped, flag rows fetched one tap ahead, per-axis membership tables, and the four waves of a
 * workgroup (8 x 8 pixels) walking their common bounding box in lock-step, one s_barrier per 64 positions;
 * 7 = the same with 2x8 tiles; 6 = 7 free-running (strip width 16); 4 = 6 with the flags fetched per live group and
 * pixel-by-pixel classification; 5 = 4 without the skip; 3 = 1x8 pixels without the skip; 8, 9, 11 = the flagged
 * accumulate on the matrix pipe (v_mfma_f32_4x4x1_16b_f32 with A = membership flags: every group / live groups only /
 * live groups with 4x4 tiles); 10 = four taps per v_mfma_f32_16x16x4_f32 (4x4 tiles, free-running, D a multiple of
 * 64, else 12 runs); 0 = four adjacent pixels per wave, 16-way switch on the mask; 1 = plain one-pixel-per-wave walk
 * (the only form for volumes >= 4 GiB, D > 256 and order 2); 2 = pipelined walk.  Every variant produces the same bits;
 * the matrix-pipe forms are measured equal to or slower than the default. */
int smt_crossarm_set_variant(smt_crossarm *h, int variant);
/* Tuning hook: width (multiple of 4) of the column strips each XCD sweeps (all variants but 1;
 * variants 3 and 4 round it to 8, 16 or a multiple of 32). */
int smt_crossarm_set_strip_width(smt_crossarm *h, int width);
/* Tuning hook: aggregation waves per SIMD (3, 4 or 5, enforced through an LDS claim per workgroup; 0 = whatever the
 * register count allows, i.e. 6: the default).  Limiting it leaves VGPRs for kernels of other streams, which on this
 * path buys nothing (the scanline passes then run beside the aggregation and both slow down).
 * SMT_AGG_WAVES in the environment overrides the default for every handle. */
int smt_crossarm_set_occupancy(smt_crossarm *h, int waves_per_simd);
/* Tuning hook (variants 3-5): 0 = column strips interleaved over the 8 XCDs, 1 = every XCD owns one
 * contiguous band of rows and sweeps it strip by strip.  Placement only; results are identical. */
int smt_crossarm_set_sweep(smt_crossarm *h, int sweep);

/* CBLSM.h:327-381 ComputeAD / ComputeADRight on uchar images -> float volume. */
int smt_cblsm_ad(const uint8_t *L, const uint8_t *R, int H, int W, int D, int view, float *vol,
                 void *stream);

/* CBLSM.h:65-236 chooseArmLengthLeft / Right / Up / Down (dir 0 / 1 / 2 / 3): per-hypothesis arm
 * lengths, int32 [H][W][D], from the two views' arm maps (int32 [H][W], e.g. smt_crossarm_arm_maps
 * of a left-image and a right-image handle).  own_arm = ArmLL / ArmLR / ArmLUp / ArmLDown;
 * other_vertical_arm = ArmRUp / ArmRDown for dir 2 / 3 (ignored for 0 / 1, may be NULL).  The
 * reference's call sites are commented out (CBLSM.cpp:108-111); the up / down forms read the right
 * view's arms at row i -/+ k for k <= own_arm, so own_arm must stay inside the image, as the arm
 * kernels guarantee. */
int smt_cblsm_choose_arm_length(int dir, const int *own_arm, const int *other_vertical_arm,
                                const int *armRL, const int *armRR, int H, int W, int D,
                                int *arm_volume, void *stream);

/* CBLSM.h:1087-1126 costAggregationNew with :969-1045 ComputeLocalValue (dead experiment, call site commented
 * out at CBLSM.cpp:113-116): cost[p][d] = | value(left image, arms at slot 0) - value(right image shifted by
 * d, arms at slot d) |, value = sum over rows [-Up, Down] of the row's pixels in [j-L-d, j+R-d) divided by
 * the sum of (L+R+1) -- the reference counts one pixel more per row than it adds (:1021, :1034), R+1 for
 * a left-clipped row (:1011) and 1 for a row clipped to column 0 (:996); all reproduced.
 *   Lp, Rp      uint8 [H+2w][W+2w], replicate-padded by w = winSize+1
 *   armvol*     int32 [H][W][D] from smt_cblsm_choose_arm_length (dir 0, 1, 2, 3); Up / Down must keep
 *               rows inside the image (rows outside are skipped)
 *   cost        float32 [H][W][D] out. */
int smt_cblsm_cost_aggregation_new(const uint8_t *Lp, const uint8_t *Rp, int H, int W, int D, int winSize,
                                   const int *armvolL, const int *armvolR, const int *armvolUp,
                                   const int *armvolDown, float *cost, void *stream);

/* =====================================================================================
 * Scanline optimiser                  replaces class ScanlineOptimizer
 *                                     (AD-CensusV1/ScanlineOptimizer.h)
 * ===================================================================================== */
typedef struct smt_scanline smt_scanline;

/* ScanlineOptimizer::Initialize (:66-79).  The reference allocates five volumes; this
 * engine keeps one scratch volume. */
int smt_scanline_create(int H, int W, int D, int p1, int p2, smt_scanline **out);
int smt_scanline_create_on(int device, int H, int W, int D, int p1, int p2, smt_scanline **out);
int smt_scanline_destroy(smt_scanline *h);
int smt_scanline_set_stream(smt_scanline *h, void *stream);

/* ScanlineOptimizer::ScanLine (:104-128): the four passes and ((left+right)+up)+down,
 * written to vol_out (`_ProcessedVolume`).  gray: float32 [H][W] guidance image
 * (`leftptr`, main.cpp:88).  If disp != NULL, ScanlineOptimizer::WTA (:40-64) is fused.
 * vol_out must not alias vol_in. */
int smt_scanline_run(smt_scanline *h, const float *vol_in, const float *gray, float *vol_out,
                     float *disp);

/* One path volume only (leftVolume/rightVolume/upVolume/downVolume), for tests.
 * pass: 0 left->right (isLeft=true), 1 right->left, 2 top->bottom (isUp=true), 3 bottom->top. */
int smt_scanline_pass(smt_scanline *h, const float *vol_in, const float *gray, int pass,
                      float *vol_out);

/* =====================================================================================
 * The whole AD-CensusV1/main.cpp pipeline, batched      (SURVEY 8b "batch variants"; BASELINE configs[2])
 * ===================================================================================== */
typedef struct smt_pipeline smt_pipeline;
typedef struct smt_pipeline_params {
    float sigmaC, sigmaS;   /* 10, 30   main.cpp:25-26 */
    int tao, p1, p2, gate;  /* 30, 10, 150, 2   main.cpp:27-30 */
} smt_pipeline_params;
void smt_pipeline_default_params(smt_pipeline_params *p);
/* Owns one AD_Census, one CrossArmAggregation, one ScanlineOptimizer and the three volumes between them. */
int smt_pipeline_create(int H, int W, int D, const smt_pipeline_params *p, smt_pipeline **out);
int smt_pipeline_create_on(int device, int H, int W, int D, const smt_pipeline_params *p, smt_pipeline **out);
int smt_pipeline_destroy(smt_pipeline *h);
int smt_pipeline_set_stream(smt_pipeline *h, void *stream);
/* main.cpp:46-92 for `pairs` pairs of uint8 gray images [pairs][H][W] (the images after cvtColor, :19-20), in
 * main.cpp's order with lines 86-89 and 92 enabled: float copies, ComputeADcensus(+Right), CrossArm
 * Initialize + four arm passes + AggregationVertical + WTA on the left and on the right image,
 * ScanlineOptimizer on the LEFT aggregated volume + WTA, LeftRightConsistency(gate).
 *   dispL   float32 [pairs][H][W]: the scanline WTA map after the LR check (+inf = rejected)
 *   dispR   float32 [pairs][H][W]: WTA of the aggregated right volume (main.cpp:84)
 *   cls     uint8   [pairs][H][W] as smt_lrcheck;  counts int32 [pairs][2] (may be NULL)
 * Asynchronous: the call enqueues and returns; part of the work runs on streams the handle owns (the right
 * view's aggregation beside the left view's scanline passes, see csrc/pipeline.hip), but everything a call
 * enqueued is ordered before whatever the caller enqueues next on the handle's stream, and after whatever the
 * caller enqueued there before the call (the inputs).  One caller thread per handle.  The volumes are reused
 * per pair (smt_pipeline_volumes: the last pair's, borrowed).  This is the sharding unit for the pair axis of
 * config 3.  SMT_PIPE_SCHEDULE=0|1|2 (environment, read at create) selects the stream schedule; results
 * are identical. */
int smt_pipeline_run_batch(smt_pipeline *h, const uint8_t *grayL, const uint8_t *grayR, int pairs,
                           float *dispL, float *dispR, uint8_t *cls, int *counts);
int smt_pipeline_volumes(smt_pipeline *h, float **cost_left, float **cost_right, float **agg_left,
                         float **agg_right, float **scanline_sum);
int smt_pipeline_status(smt_pipeline *h); /* synchronising: SMT_ERR_DOMAIN / SMT_ERR_REF_UB seen since the last call */

/* =====================================================================================
 * Left-right consistency              replaces LeftRightConsistency
 *                                     (AD-CensusV1/PostProcessing.h:72-135)
 * ===================================================================================== */
/* In place on dispL (invalid -> +inf).  cls: uint8 [H][W], 0 kept / 1 occlusion /
 * 2 mismatch; the reference's two vectors are these classes in row-major order
 * (smt_lrcheck_lists rebuilds them on the host).  counts: device int32[2] = {occlusions,
 * mismatches}, may be NULL. */
int smt_lrcheck(float *dispL, const float *dispR, int H, int W, int gate, uint8_t *cls,
                int *counts, void *stream);

/* LeftAndRightConsistency (AD-CensusV1/PostProcessing.h:10-70; no call site): the out-of-place sibling.
 * dispL is only read; lastDisp float32 [H][W] receives dispL where the pixel is kept and 0 where it is
 * rejected; the test is abs(d - dR) >= gate with a float gate (:32), no +inf pre-check.  Classes as
 * smt_lrcheck.  A disparity for which `static_cast<int>(j - disp + 0.5)` overflows int (non-finite or
 * huge; undefined in C++) is treated as x86 does: INT_MIN, i.e. out of range -> mismatch. */
int smt_lrcheck_variant(const float *dispL, const float *dispR, float *lastDisp, int H, int W, float gate,
                        uint8_t *cls, int *counts, void *stream);

/* Host helper: expand a HOST copy of cls into the reference's (row, col) pair lists.
 * Each list must have room for H*W pairs (2 ints per pair); returns counts. */
int smt_lrcheck_lists(const uint8_t *cls_host, int H, int W, int *occlusion_pairs, int *n_occ,
                      int *mismatch_pairs, int *n_mis);

/* FillTheHole (AD-CensusV1/PostProcessing.h:156-248; same text in CBLSM/PostProcessing.h).
 * In place on the DEVICE map disp (row*col floats).  The reference swaps the extents
 * (`width = row`, `height = col`, :158-159) and this is reproduced: the buffer is addressed as
 * `col` lines of `row` entries, and a hole is an entry equal to 65535.0f (:182, :212) -- not the
 * +inf that LeftRightConsistency writes.  occ / mis: HOST arrays of (first, second) int pairs in
 * list order, as LeftRightConsistency fills them.  Pass 0 gives each occlusion the second
 * smallest of the first non-hole values met along 8 rays, pass 1 each mismatch their median,
 * pass 2 (only when the mismatch list is not empty, :174) every remaining hole the median.
 * The reference then leaves the third pass's pixel list in the caller's `mismatch` vector
 * (:186): third (HOST, room for row*col pairs, may be NULL) and *n_third (may be NULL; -1 when
 * the list was not replaced) return it.  SMT_ERR_REF_UB where the reference writes out of
 * bounds: a listed pair outside the buffer (nothing is modified), or more third-pass holes than
 * the mismatch list had entries (`fill_disps` is sized before the list is replaced, :177 vs
 * :186; passes 0 and 1 have been applied by then, as in the reference).  Synchronising. */
int smt_fill_the_hole(float *disp, int row, int col, int dispRange, const int *occlusion_pairs,
                      int n_occ, const int *mismatch_pairs, int n_mis, int *third_pairs,
                      int *n_third, void *stream);

/* =====================================================================================
 * CrossAggregator (vendored ethan-li AD-Census)   replaces class CrossAggregator
 *                                     (CBLSM/cross_aggregator.{h,cpp})
 * ===================================================================================== */
typedef struct smt_crossagg smt_crossagg;

/* Initialize(width,height,min_disparity,max_disparity) (:19-58). D = max-min.
 * Returns SMT_ERR_ARG where the reference returns false. */
int smt_crossagg_create(int W, int H, int D, smt_crossagg **out);
int smt_crossagg_create_on(int device, int W, int H, int D, smt_crossagg **out);
int smt_crossagg_destroy(smt_crossagg *h);
int smt_crossagg_set_stream(smt_crossagg *h, void *stream);
/* SetParams (:67-74); defaults L1=34 L2=17 t1=20 t2=6 (adcensus_types.h:69-70). */
int smt_crossagg_set_params(smt_crossagg *h, int L1, int L2, int t1, int t2);
/* SetData + Aggregate(num_iters) (:60-65, :89-118).  img_left: uint8 [H][W][3];
 * cost_init: float32 [H][W][D].  The result stays in the handle (get_cost_ptr). */
int smt_crossagg_aggregate(smt_crossagg *h, const uint8_t *img_left, const float *cost_init,
                           int num_iters);
/* Test hook: 2 = shared-tap passes (16 pixels per wave along the pass axis, default), 1 = one pixel per wave
 * (first formulation).  Identical bits. */
int smt_crossagg_set_impl(smt_crossagg *h, int impl);
/* get_cost_ptr (:125-133) / get_arms_ptr (:120-123): borrowed. arms: uint8 [H][W][4] =
 * left,right,top,bottom (struct CrossArm, cross_aggregator.h:17-20). */
int smt_crossagg_cost(smt_crossagg *h, float **cost);
int smt_crossagg_arms(smt_crossagg *h, uint8_t **arms);

/* ADCensusOption (CBLSM/adcensus_types.h:45-75), field for field, with its constructor's defaults
 * (smt_adcensus_option_default).  SURVEY 8f n3.  The reference tree holds this struct and the aggregator it
 * feeds but NOT the rest of ethan-li-coding/AD-Census (cost computer, its own scanline optimiser with
 * so_p1 / so_p2 / so_tso, the multi-step refiner with irv_ts / irv_th, filling, discontinuity adjustment,
 * sub-pixel): those fields are carried, nothing consumes them, and no part of that flow is claimed here
 * (it would be "parity unpinned" against a source that is not in /root/reference). */
typedef struct smt_adcensus_option {
    int32_t min_disparity, max_disparity;
    int32_t lambda_ad, lambda_census;
    int32_t cross_L1, cross_L2, cross_t1, cross_t2;
    float so_p1, so_p2;
    int32_t so_tso, irv_ts;
    float irv_th, lrcheck_thres;
    int32_t do_lr_check, do_filling, do_discontinuity_adjustment;      /* bool in the reference */
} smt_adcensus_option;
void smt_adcensus_option_default(smt_adcensus_option *o);
/* The one caller shape the reference holds for it (CBLSM/CBLSM.cpp:138-143, commented out; WTA :152):
 *   CrossAggregator a; a.Initialize(col, row, 0, dispRange); a.SetData(bytes_left, bytes_right, dispVolum);
 *   a.SetParams(option.cross_L1, option.cross_L2, option.cross_t1, option.cross_t2); a.Aggregate(4);
 *   cost = a.get_cost_ptr();  ComputeDispOringin(cost, disp, ...)
 * in one call: D = max_disparity - min_disparity, bytes_left uint8 [H][W][3], cost_init / cost_out float32
 * [H][W][D], disp float32 [H][W] (may be NULL).  Synchronising. */
int smt_adcensus_option_aggregate(const smt_adcensus_option *o, const uint8_t *bytes_left, const float *cost_init,
                                  int W, int H, int num_iters, float *cost_out, float *disp, void *stream);

/* =====================================================================================
 * Window matchers                     replace SAD/Sad.h, NCC/NCC.h, ASW/ASW.h
 * ===================================================================================== */
/* GetPointDepthLeft (Sad.h:96-139, view SMT_VIEW_LEFT, WTA = OptimalDisparity :40-85) /
 * GetPointDepthRight (:141-182, view SMT_VIEW_RIGHT, WTA = GetMinSadIndex :22-38).
 * Lp, Rp: uint8 [H+2w][W+2w] replicate-padded by w = winsize+1 (SADmain.cpp:47-48);
 * window side 2w+1.  disp: int32 [H][W] (right view leaves the last row/column 0). */
int smt_sad(const uint8_t *Lp, const uint8_t *Rp, int H, int W, int D, int winsize, int view,
            int32_t *disp, void *stream);
/* Test hook (process-wide): 2 = window rows staged in LDS as byte-shifted dword copies, 32 pixels per workgroup
 * (default for windows of side >= 4), 1 = one wave per pixel straight from global memory (first formulation, and the
 * fallback for 3x3 windows).  Identical results. */
int smt_sad_set_impl(int impl);
/* CrossCheckDiaparity (Sad.h:184-222).  out int32 [H][W] (invalid = INT32_MIN, the x86
 * value of the reference's int(inf)); cls as smt_lrcheck. */
int smt_sad_crosscheck(const int32_t *dispL, const int32_t *dispR, int H, int W, int32_t *out,
                       uint8_t *cls, void *stream);

/* NCC_algorithem (NCC.h:69-95) = ComputeCost (:15-49, float64) + WinTakeAll (:53-67,
 * argMAX with a float32-narrowed running maximum).  L, R uint8 [H][W] unpadded; only the
 * interior winSize <= i < H-winSize, winSize <= j < W-winSize is written, the rest of
 * disp is set to 0.  cost (optional, may be NULL): float64 [H][W][D] per-hypothesis
 * costs for tolerance checks. */
int smt_ncc(const uint8_t *L, const uint8_t *R, int H, int W, int D, int winSize, int32_t *disp,
            double *cost, void *stream);
/* Test hook (process-wide): 2 = window statistics once per image + the cross term by v_dot4_u32_u8 (default for
 * windows up to 31x31; needs 24*H*W bytes of stream-ordered scratch for the duration of the call), 1 = the
 * reference's loop nest, one lane per hypothesis (also the fallback).  The two agree to ~1e-14 relative on the
 * costs (both within the 1e-4 tolerance of the reference's own rounding) and give the same NaN pattern. */
int smt_ncc_set_impl(int impl);

/* getGausssianMask (ASW.h:16-35) and getColorMask (:41-47), computed on the HOST in
 * float64 exactly as the reference does.  space: (2*winSize+3)^2 doubles, color: 256. */
int smt_asw_masks(int winSize, double sigma_space, double sigma_color, double *space_host,
                  double *color_host);

/* AdaptiveSupportWeight (ASW.h:329-378, SMT_VIEW_LEFT) / AdaptiveSupportWeightRight
 * (:382-431, SMT_VIEW_RIGHT): bilateralfiterWight (:210-257) per hypothesis + WinTakeAll
 * (:193-208).  Lp, Rp: uint8 [H+2w][W+2w] replicate-padded by w = winSize+1
 * (ASWeight.cpp:54-57); space/color: DEVICE float64 tables from smt_asw_masks; T: error
 * truncation.  disp float32 [H][W]; cost (optional) float32 [H][W][D]. */
int smt_asw(const uint8_t *Lp, const uint8_t *Rp, int H, int W, int D, int winSize,
            const double *space, const double *color, int T, int view, float *disp, float *cost,
            void *stream);
/* Scratch device memory of smt_asw / smt_ncc comes from an arena the library owns (csrc/scratch.hip: hipMalloc'ed
 * blocks cached per device and handed out stream-ordered on the caller's stream).  The arena keeps what it has grown
 * to -- the anchor weights of an smt_asw call (H*W*(2*winSize+3)^2*8 bytes while that is under 6 GiB, beyond it one
 * slot per workgroup in flight: 160 MB at 35 x 35 whatever the image size), 24*H*W bytes after an smt_ncc -- until the
 * process ends or the host asks
 * for it back: smt_scratch_trim synchronises the current device and returns every idle block beyond `keep_bytes` to
 * the driver (hipFree); smt_scratch_info reports what the arena holds / has handed out.
 * Why not hipMallocAsync: on ROCm 7.2 a stream-ordered pool that trims and grows again hands out a block that is
 * zero-filled while the kernels already run on it (wrong ASW maps in round 2; tools/asw_bisect.py).
 * SMT_SCRATCH_MODE=pool|default|malloc (environment, for that tool) selects a never-trimming hipMemPool / the
 * device's default pool (the failing configuration) / plain hipMalloc per call. */
int smt_scratch_trim(size_t keep_bytes);
int smt_scratch_info(size_t *reserved_bytes, size_t *used_bytes);
/* Test hook (process-wide): which ASW formulation runs.  0 (default) = 3 while the whole-image anchor table stays
 * under 6 GiB (SMT_ASW_TABLE_MAX_MB), 6 beyond.  3 = per-row other-image weight tables in LDS, anchor weights from a
 * whole-image table written by a table kernel first (H*W*(2*winSize+3)^2*8 bytes of scratch: 5 GB at 960x540, 35x35),
 * two pixels per wave; 6 = the same tap loop with the anchor weights in one scratch slot per workgroup in flight, rebuilt
 * by the workgroup for every tile it takes (k_asw4: 160 MB at 35x35 whatever the image size, 7 % slower at config 4);
 * 4 = 3 with one pixel per wave; 5 = 3 with the anchor operands read by vector loads instead of through the scalar
 * cache; 1 = the first formulation (everything recomputed per tap; also the fallback when the scratch cannot be had).
 * All produce identical bits. */
int smt_asw_set_impl(int impl);
/* Batch variants of the three window matchers ("batch variants taking a pair count and strides"):
 * `pairs` image pairs and maps, consecutive pairs `img_stride` / `disp_stride` ELEMENTS apart (0 = dense: one
 * padded image, (H+2w)*(W+2w) bytes for SAD / ASW and H*W for NCC; one map, H*W).  Exactly the results of
 * `pairs` single calls, enqueued on `stream`; the per-call scratch of NCC / ASW is reused from pair to pair. */
int smt_sad_batch(const uint8_t *Lp, const uint8_t *Rp, int pairs, size_t img_stride, int H, int W, int D, int winsize,
                  int view, int32_t *disp, size_t disp_stride, void *stream);
int smt_ncc_batch(const uint8_t *L, const uint8_t *R, int pairs, size_t img_stride, int H, int W, int D, int winSize,
                  int32_t *disp, size_t disp_stride, void *stream);
int smt_asw_batch(const uint8_t *Lp, const uint8_t *Rp, int pairs, size_t img_stride, int H, int W, int D, int winSize,
                  const double *space, const double *color, int T, int view, float *disp, size_t disp_stride,
                  void *stream);
/* CrossCheckDiaparity (ASW.h:108-145): float maps -> uint8 map, 0 = rejected. */
int smt_asw_crosscheck(const float *dispL, const float *dispR, int H, int W, uint8_t *out,
                       void *stream);

/* =====================================================================================
 * Either side of the path: input staging and the first post-filter
 * ===================================================================================== */
/* Image files, HOST side (no GPU work): what the reference's drivers do with cv::imread(path) /
 * cv::imwrite(path, img) (AD-CensusV1/main.cpp:16-17, :115-117; SADmain.cpp:28-29; ASWeight.cpp:11-12),
 * without OpenCV / libpng / zlib.
 * smt_image_read: 8-bit PNG (colour types 0, 2, 3, 4, 6; bit depths 1-16 -- 16-bit samples keep their
 * high byte; alpha dropped; palettes expanded; non-interlaced) and binary PGM / PPM (P5 / P6).
 *   want_channels  3: always 3-channel B, G, R like cv::imread's default flag (a gray file is replicated);
 *                  1: gray (a colour file goes through the BGR2GRAY rule of smt_bgr2gray);
 *                  0: as stored (1 or 3).
 *   *pixels is malloc'd [H][W][channels]; release it with smt_image_free.
 * smt_image_write: by extension -- .png (8-bit gray or colour, filter 0, stored deflate blocks: valid,
 * uncompressed), .pgm / .ppm / .pnm; channels 1 or 3 (B, G, R in memory).
 * Both return SMT_ERR_ARG for unreadable / malformed / unsupported files. */
int smt_image_read(const char *path, int want_channels, uint8_t **pixels, int *H, int *W, int *channels);
int smt_image_free(uint8_t *pixels);
int smt_image_write(const char *path, const uint8_t *pixels, int H, int W, int channels);

/* cvtColor(CV_BGR2GRAY) as the drivers call it (AD-CensusV1/main.cpp:19-20): OpenCV 3.1.0's
 * 8-bit fixed-point rule (1868 B + 9617 G + 4899 R + 8192) >> 14.  bgr uint8 [H][W][3]. */
int smt_bgr2gray(const uint8_t *bgr, int H, int W, uint8_t *gray, void *stream);
/* copyMakeBorder(..., BORDER_REPLICATE) with equal borders (SADmain.cpp:47-48, ASWeight.cpp:54-57).
 * dst uint8 [H+2*pad][W+2*pad]. */
int smt_pad_replicate(const uint8_t *src, int H, int W, int pad, uint8_t *dst, void *stream);
/* uchar -> float image copy (main.cpp:46-55). */
int smt_u8_to_f32(const uint8_t *src, int H, int W, float *dst, void *stream);
/* MedianFilter(in, out, width, height, wnd_size) (AD-CensusV1/PostProcessing.h:314-344): median of
 * the in-image part of the window, element [n/2] of the ascending order.  wnd_size <= 7. */
int smt_median_filter(const float *in, float *out, int W, int H, int wnd_size, void *stream);
/* RemoveSpeckles(disparity_map, width, height, diff_insame, min_speckle_aera, invalid_val)
 * (AD-CensusV1/PostProcessing.h:250-311), in place.  invalid_val is an int as in the reference
 * (its call sites pass +inf: undefined conversion, INT_MIN on x86).  Synchronising. */
int smt_remove_speckles(float *disparity_map, int W, int H, int diff_insame, unsigned min_speckle_area,
                        int invalid_val, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SMT_H_ */
