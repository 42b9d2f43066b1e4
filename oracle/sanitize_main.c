/* AddressSanitizer / UBSan run of the CPU oracle (the reference has real races and
 * undefined conversions; the restatement must be clean).  Built by `make -C oracle sanitize` as one
 * translation unit with smt_oracle.c, every stage called on small heap buffers of exactly the documented
 * sizes, so any out-of-bounds access, signed overflow, misaligned or invalid conversion aborts the run.
 * TEST INFRASTRUCTURE ONLY. */
#include "smt_oracle.c"
#include <stdio.h>

#define NEW(T, n) ((T *)malloc(sizeof(T) * (size_t)(n)))

int main(void)
{
    uint64_t acc = 0;
    for (int cfg = 0; cfg < 3; cfg++) {
        const int H = cfg == 0 ? 24 : cfg == 1 ? 17 : 9, W = cfg == 0 ? 52 : cfg == 1 ? 40 : 33, D = cfg == 0 ? 20 : cfg == 1 ? 7 : 70;
        const size_t n = (size_t)H * W, V = n * D;
        uint8_t *L8 = NEW(uint8_t, n), *R8 = NEW(uint8_t, n);
        orc_synth_pair(H, W, D, 11 + cfg, cfg == 1, L8, R8);
        float *Lf = NEW(float, n), *Rf = NEW(float, n);
        for (size_t k = 0; k < n; k++) { Lf[k] = L8[k]; Rf[k] = R8[k]; }
        float *cl = NEW(float, V), *cr = NEW(float, V), *dl = NEW(float, n), *dr = NEW(float, n);
        if (orc_adcensus_view(Lf, Rf, H, W, D, 10.f, 30.f, 0, 0, H, cl) || orc_adcensus_view(Lf, Rf, H, W, D, 10.f, 30.f, 1, 0, H, cr)) return 2;
        orc_wta(cl, H, W, D, dl); orc_wta(cr, H, W, D, dr);
        int *a[4];
        for (int k = 0; k < 4; k++) a[k] = NEW(int, n);
        float *ag = NEW(float, V), *so = NEW(float, V);
        if (orc_arms_all(L8, H, W, 1, 30, 6, 17, 34, 1, 0, a[0], a[1], a[2], a[3])) return 3;
        for (int order = 0; order < 3; order++) (void)orc_aggregate_rect(cl, H, W, D, a[0], a[1], a[2], a[3], order, ag);
        if (orc_arms_all(L8, H, W, 1, 30, 6, 17, 34, 1, 1, a[0], a[1], a[2], a[3]) == 0)
            (void)orc_aggregate_rect(cl, H, W, D, a[0], a[1], a[2], a[3], 0, ag);   /* stride-bug arms: out-of-plane taps are counted, not read */
        orc_arms_all(L8, H, W, 1, 25, 6, 17, 34, 0, 0, a[0], a[1], a[2], a[3]);
        (void)orc_aggregate_rect(cl, H, W, D, a[0], a[1], a[2], a[3], 1, ag);
        if (orc_scanline(ag, Lf, H, W, D, 10, 150, so)) return 4;
        orc_wta(so, H, W, D, dl);
        uint8_t *cls = NEW(uint8_t, n);
        long no, nm;
        float *last = NEW(float, n);
        dl[3] = INFINITY; dl[5] = NAN; dl[7] = -INFINITY; dl[9] = 4e9f; dr[4] = -5e9f;     /* the UB conversions */
        orc_lrcheck_variant(dl, dr, last, H, W, 1.5f, cls, &no, &nm);
        orc_lrcheck(dl, dr, H, W, 2, cls, &no, &nm);
        acc ^= orc_fnv1a(so, V * 4) ^ orc_fnv1a(dl, n * 4) ^ (uint64_t)no;
        /* CBLSM pieces */
        int *al[4], *ar[4], *vol[4];
        for (int k = 0; k < 4; k++) { al[k] = NEW(int, n); ar[k] = NEW(int, n); vol[k] = NEW(int, V); }
        orc_arms_all(L8, H, W, 1, 25, 6, 17, 34, 0, 0, al[0], al[1], al[2], al[3]);
        orc_arms_all(R8, H, W, 1, 25, 6, 17, 34, 0, 0, ar[0], ar[1], ar[2], ar[3]);
        orc_choose_arm_length(0, al[0], NULL, ar[0], ar[1], H, W, D, vol[0]);
        orc_choose_arm_length(1, al[1], NULL, ar[0], ar[1], H, W, D, vol[1]);
        orc_choose_arm_length(2, al[2], ar[2], ar[0], ar[1], H, W, D, vol[2]);
        orc_choose_arm_length(3, al[3], ar[3], ar[0], ar[1], H, W, D, vol[3]);
        orc_cblsm_ad(L8, R8, H, W, D, 0, ag); orc_cblsm_ad(L8, R8, H, W, D, 1, ag);
        {
            const int win = 1, w = win + 1, Hp = H + 2 * w, Wp = W + 2 * w;
            uint8_t *Lp = NEW(uint8_t, (size_t)Hp * Wp), *Rp = NEW(uint8_t, (size_t)Hp * Wp);
            for (int i = 0; i < Hp; i++)
                for (int j = 0; j < Wp; j++) {
                    int ii = i - w < 0 ? 0 : i - w >= H ? H - 1 : i - w, jj = j - w < 0 ? 0 : j - w >= W ? W - 1 : j - w;
                    Lp[(size_t)i * Wp + j] = L8[(size_t)ii * W + jj]; Rp[(size_t)i * Wp + j] = R8[(size_t)ii * W + jj];
                }
            orc_cblsm_cost_aggregation_new(Lp, Rp, Hp, Wp, win, vol[0], vol[1], vol[2], vol[3], D, ag);
            acc ^= orc_fnv1a(ag, V * 4);
            /* window matchers on the same padded pair */
            int32_t *sd = (int32_t *)calloc(n, sizeof(int32_t)), *sd2 = (int32_t *)calloc(n, sizeof(int32_t)), *so2 = NEW(int32_t, n);
            orc_sad(Lp, Rp, Hp, Wp, D, win, 0, sd); orc_sad(Lp, Rp, Hp, Wp, D, win, 1, sd2);
            orc_sad_crosscheck(sd, sd2, H, W, so2, cls);
            double *sp = NEW(double, (2 * win + 3) * (2 * win + 3)), cm[256];
            orc_asw_masks(win, 50.0, 30.0, sp, cm);
            float *ad = NEW(float, n), *ad2 = NEW(float, n), *ac = NEW(float, V);
            orc_asw(Lp, Rp, Hp, Wp, D, win, sp, cm, 40, 0, 0, H, ad, ac);
            orc_asw(Lp, Rp, Hp, Wp, D, win, sp, cm, 40, 1, 0, H, ad2, NULL);
            orc_asw_crosscheck(ad, ad2, H, W, cls);
            float *med = NEW(float, n);
            orc_median(ad, med, W, H, 3);
            orc_remove_speckles(med, W, H, 1, 20, -2147483647 - 1);
            acc ^= orc_fnv1a(sd, n * 4) ^ orc_fnv1a(med, n * 4);
            free(Lp); free(Rp); free(sd); free(sd2); free(so2); free(sp); free(ad); free(ad2); free(ac); free(med);
        }
        {
            int32_t *nd = (int32_t *)calloc(n, sizeof(int32_t));
            double *nc = NEW(double, V);
            if (H > 6 && W > 6) orc_ncc(L8, R8, H, W, D, 2, 0, H, nd, nc);
            acc ^= orc_fnv1a(nd, n * 4);
            free(nd); free(nc);
        }
        {   /* CrossAggregator */
            uint8_t *bgr = NEW(uint8_t, n * 3), *arms = NEW(uint8_t, n * 4), *g = NEW(uint8_t, n);
            for (size_t k = 0; k < n; k++) { bgr[3 * k] = L8[k]; bgr[3 * k + 1] = (uint8_t)(L8[k] ^ 3); bgr[3 * k + 2] = R8[k]; }
            if (orc_crossagg(bgr, cl, W, H, D, 34, 17, 20, 6, 4, arms, ag)) return 5;
            orc_bgr2gray(bgr, (int)n, g);
            acc ^= orc_fnv1a(ag, V * 4) ^ orc_fnv1a(g, n);
            free(bgr); free(arms); free(g);
        }
        {   /* FillTheHole on an LR-checked map */
            long cap = (long)n;
            int *occ = NEW(int, 2 * cap), *mis = NEW(int, 2 * cap), *third = NEW(int, 2 * cap), nt = -1, nocc = 0, nmis = 0;
            for (int i = 0; i < H; i++)
                for (int j = 0; j < W; j++) {
                    if (cls[(size_t)i * W + j] == 1) { occ[2 * nocc] = i; occ[2 * nocc + 1] = j; nocc++; }
                    else if (cls[(size_t)i * W + j] == 2) { mis[2 * nmis] = i; mis[2 * nmis + 1] = j; nmis++; }
                }
            for (size_t k = 0; k < n; k++) if (!isfinite(dl[k])) dl[k] = 65535.0f;
            (void)orc_fill_the_hole(dl, H, W, D, occ, nocc, mis, nmis, third, &nt);
            free(occ); free(mis); free(third);
        }
        for (int k = 0; k < 4; k++) { free(a[k]); free(al[k]); free(ar[k]); free(vol[k]); }
        free(L8); free(R8); free(Lf); free(Rf); free(cl); free(cr); free(dl); free(dr); free(ag); free(so); free(cls); free(last);
    }
    printf("oracle sanitizer run clean, checksum %016llx\n", (unsigned long long)acc);
    return 0;
}
