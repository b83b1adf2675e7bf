/* Sanitizer driver of the CPU oracle (test infrastructure): runs orc_match with every tap on small
 * pseudo-random frames of awkward shapes under ASan + UBSan (make -C oracle sanitize). Prints one
 * checksum line per case so that a run can also be compared with the optimised library's output. */
#include "adcensus_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t rng_state = 12345u;
static uint32_t rng(void) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5;
    return rng_state;
}

static int run_case(int W, int H, int dmin, int dmax, int l1) {
    const size_t npx = (size_t)W * H, D = (size_t)(dmax - dmin);
    uint8_t *L = malloc(npx * 3), *R = malloc(npx * 3);
    /* blocky texture so that arms, occlusions and mismatches all occur */
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < 3; c++) {
                const uint32_t base = (uint32_t)((x / 5) * 37 + (y / 4) * 91 + c * 50) & 255u;
                L[((size_t)y * W + x) * 3 + c] = (uint8_t)((base + (rng() & 7u)) & 255u);
                const int xs = x + 2 + dmin < W && x + 2 + dmin >= 0 ? x + 2 + dmin : x;
                const uint32_t b2 = (uint32_t)((xs / 5) * 37 + (y / 4) * 91 + c * 50) & 255u;
                R[((size_t)y * W + x) * 3 + c] = (uint8_t)((b2 + (rng() & 7u)) & 255u);
            }
    orc_option o;
    orc_default_option(&o);
    o.min_disparity = dmin;
    o.max_disparity = dmax;
    o.cross_L1 = l1;
    orc_taps t;
    memset(&t, 0, sizeof(t));
    t.gray_left = malloc(npx); t.gray_right = malloc(npx);
    t.census_left = malloc(npx * 8); t.census_right = malloc(npx * 8);
    t.cost_init = malloc(npx * D * 4); t.cost_aggr = malloc(npx * D * 4);
    t.cost_scan = malloc(npx * D * 4); t.cost_scan_aux = malloc(npx * D * 4);
    t.arms = malloc(npx * 4);
    t.sup_count_h = malloc(npx * 2); t.sup_count_v = malloc(npx * 2);
    t.disp_left_wta = malloc(npx * 4); t.disp_right_wta = malloc(npx * 4);
    t.disp_outlier = malloc(npx * 4); t.disp_voting = malloc(npx * 4);
    t.disp_interp = malloc(npx * 4); t.disp_edge = malloc(npx * 4);
    t.mask_occ = malloc(npx); t.mask_mis = malloc(npx);
    t.mask_occ_voting = malloc(npx); t.mask_mis_voting = malloc(npx);
    t.edge_mask = malloc(npx);
    float *disp = malloc(npx * 4);
    const int rc = orc_match(L, R, W, H, &o, disp, &t);
    uint32_t sum = 0;
    for (size_t i = 0; i < npx; i++) { uint32_t b; memcpy(&b, &disp[i], 4); sum = sum * 31u + b; }
    printf("%dx%d d[%d,%d) L1=%d rc=%d checksum=%08x\n", W, H, dmin, dmax, l1, rc, sum);
    free(L); free(R); free(disp);
    free(t.gray_left); free(t.gray_right); free(t.census_left); free(t.census_right);
    free(t.cost_init); free(t.cost_aggr); free(t.cost_scan); free(t.cost_scan_aux); free(t.arms);
    free(t.sup_count_h); free(t.sup_count_v); free(t.disp_left_wta); free(t.disp_right_wta);
    free(t.disp_outlier); free(t.disp_voting); free(t.disp_interp); free(t.disp_edge);
    free(t.mask_occ); free(t.mask_mis); free(t.mask_occ_voting); free(t.mask_mis_voting); free(t.edge_mask);
    return rc;
}

int main(void) {
    int bad = 0;
    bad |= run_case(61, 37, 0, 24, 34);
    bad |= run_case(33, 29, 3, 20, 9);
    bad |= run_case(40, 21, -4, 12, 34);
    bad |= run_case(7, 5, 0, 1, 34);
    bad |= run_case(1, 1, 0, 3, 34);
    bad |= run_case(23, 1, 0, 5, 40);
    bad |= run_case(1, 19, 0, 2, 34);
    return bad ? 1 : 0;
}
