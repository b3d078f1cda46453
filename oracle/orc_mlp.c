/* ORACLE -- test infrastructure, not product code.
 *
 * CPU restatement of the reference's network query: NeRF.forward (network.py:46-69), the D-NeRF
 * canonical/deformation nets (vanilla_dynamic/network.py:60-97, :281-341) and run_network
 * (render.py:31-56).  Plain C; points are processed in blocks of PB with the block as the
 * innermost (vectorisable) loop so that each output's k-sum stays a sequential f32 chain.
 * Parity with the reference is a tolerance (1e-4 on rendered values), pinned by
 * tests/golden/*.npz; only tests/, smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PB 64
#define ORC_MAX_D 16

typedef struct {
    int32_t D, W, input_ch, input_ch_views, output_ch, use_viewdirs, skips_mask, pad_;
    const float *pts_w[ORC_MAX_D];
    const float *pts_b[ORC_MAX_D];
    const float *views_w, *views_b, *feature_w, *feature_b, *alpha_w, *alpha_b, *rgb_w, *rgb_b;
    const float *out_w, *out_b; /* output_linear when !use_viewdirs; for a deformation net: _time_out (W -> 3) */
} orc_net_t;

/* y[o][p] = b[o] + sum_k w[o][k] x[k][p]  on a [*, PB] block (channel-major) */
static void lin_block(const float *w, const float *b, int out_f, int in_f, const float *x, float *y, int relu) {
    for (int o = 0; o < out_f; ++o) {
        float acc[PB];
        float bo = b ? b[o] : 0.f;
        for (int p = 0; p < PB; ++p) acc[p] = 0.f;
        const float *wr = w + (int64_t)o * in_f;
        for (int k = 0; k < in_f; ++k) {
            float wk = wr[k];
            const float *xk = x + k * PB;
            for (int p = 0; p < PB; ++p) acc[p] += wk * xk[p];
        }
        /* nn.Linear adds the bias after the product (addmm) */
        for (int p = 0; p < PB; ++p) {
            float v = acc[p] + bo;
            y[o * PB + p] = (relu && v < 0.f) ? 0.f : v;
        }
    }
}

/* trunk: pts_linears with skip concat (network.py:48-53). in: xin [input_ch][PB]; out: h [W (+input_ch)][PB];
 * returns number of channels in h (W, or W+input_ch if the last layer is a skip layer). */
static int trunk_block(const orc_net_t *n, const float *xin, float *h, float *tmp) {
    int in_f = n->input_ch;
    const float *cur = xin;
    for (int i = 0; i < n->D; ++i) {
        lin_block(n->pts_w[i], n->pts_b[i], n->W, in_f, cur, tmp, 1);
        if (n->skips_mask & (1 << i)) {
            memcpy(h, xin, sizeof(float) * n->input_ch * PB);
            memcpy(h + n->input_ch * PB, tmp, sizeof(float) * n->W * PB);
            in_f = n->W + n->input_ch;
        } else {
            memcpy(h, tmp, sizeof(float) * n->W * PB);
            in_f = n->W;
        }
        cur = h;
    }
    return in_f;
}

/* NeRF.forward, network.py:46-69.  x [P, input_ch+input_ch_views] -> out [P, 4 | output_ch]. */
void orc_nerf_forward(const orc_net_t *n, const float *x, int64_t P, float *out) {
    int xin_ch = n->input_ch + n->input_ch_views;
    int oc = n->use_viewdirs ? 4 : n->output_ch;
    int64_t nblk = (P + PB - 1) / PB;
#pragma omp parallel
    {
        int hw = n->W + n->input_ch;
        if (hw < n->W + n->input_ch_views) hw = n->W + n->input_ch_views;
        float *xin = (float *)malloc(sizeof(float) * PB * (xin_ch + 3 * hw + 8));
        float *h = xin + PB * xin_ch, *tmp = h + PB * hw, *h2 = tmp + PB * hw, *o4 = h2 + PB * hw;
#pragma omp for schedule(dynamic, 4)
        for (int64_t blk = 0; blk < nblk; ++blk) {
            int64_t p0 = blk * PB;
            int np = (int)(P - p0 < PB ? P - p0 : PB);
            for (int c = 0; c < xin_ch; ++c)
                for (int p = 0; p < PB; ++p) xin[c * PB + p] = p < np ? x[(p0 + p) * xin_ch + c] : 0.f;
            int hc = trunk_block(n, xin, h, tmp);
            if (n->use_viewdirs) {
                lin_block(n->alpha_w, n->alpha_b, 1, hc, h, o4 + 3 * PB, 0);                      /* :56 */
                lin_block(n->feature_w, n->feature_b, n->W, hc, h, h2, 0);                       /* :57 */
                memcpy(h2 + n->W * PB, xin + n->input_ch * PB, sizeof(float) * n->input_ch_views * PB); /* :58 */
                lin_block(n->views_w, n->views_b, n->W / 2, n->W + n->input_ch_views, h2, tmp, 1); /* :60-62 */
                lin_block(n->rgb_w, n->rgb_b, 3, n->W / 2, tmp, o4, 0);                          /* :64 */
            } else {
                lin_block(n->out_w, n->out_b, oc, hc, h, o4, 0);                                  /* :67 */
            }
            for (int p = 0; p < np; ++p)
                for (int c = 0; c < oc; ++c) out[(p0 + p) * oc + c] = o4[c * PB + p];
        }
        free(xin);
    }
}

/* DNeRF_Deformation._query_time, vanilla_dynamic/network.py:281-305.
 * xe [P, input_ch] embedded points, te [P, input_ch_time] embedded time -> dx [P,3].
 * `n` describes the time net: input_ch = position channels (the skip re-feeds ONLY those, :303),
 * input_ch_views = time channels (first layer in = input_ch+input_ch_views), out_w/out_b = _time_out. */
void orc_time_forward(const orc_net_t *n, const float *xe, const float *te, int64_t P, float *dx) {
    int ic = n->input_ch, tc = n->input_ch_views;
    int64_t nblk = (P + PB - 1) / PB;
#pragma omp parallel
    {
        int hw = n->W + ic + tc;
        float *xin = (float *)malloc(sizeof(float) * PB * (ic + tc + 2 * hw + 4));
        float *h = xin + PB * (ic + tc), *tmp = h + PB * hw, *o3 = tmp + PB * hw;
#pragma omp for schedule(dynamic, 4)
        for (int64_t blk = 0; blk < nblk; ++blk) {
            int64_t p0 = blk * PB;
            int np = (int)(P - p0 < PB ? P - p0 : PB);
            for (int p = 0; p < PB; ++p) {
                for (int c = 0; c < ic; ++c) xin[c * PB + p] = p < np ? xe[(p0 + p) * ic + c] : 0.f;
                for (int c = 0; c < tc; ++c) xin[(ic + c) * PB + p] = p < np ? te[(p0 + p) * tc + c] : 0.f;
            }
            int in_f = ic + tc;
            const float *cur = xin;
            for (int i = 0; i < n->D; ++i) {
                lin_block(n->pts_w[i], n->pts_b[i], n->W, in_f, cur, tmp, 1);
                if (n->skips_mask & (1 << i)) {
                    memcpy(h, xin, sizeof(float) * ic * PB);
                    memcpy(h + ic * PB, tmp, sizeof(float) * n->W * PB);
                    in_f = n->W + ic;
                } else {
                    memcpy(h, tmp, sizeof(float) * n->W * PB);
                    in_f = n->W;
                }
                cur = h;
            }
            lin_block(n->out_w, n->out_b, 3, in_f, h, o3, 0);
            for (int p = 0; p < np; ++p)
                for (int c = 0; c < 3; ++c) dx[(p0 + p) * 3 + c] = o3[c * PB + p];
        }
        free(xin);
    }
}
