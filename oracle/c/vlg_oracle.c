/*
 * TEST INFRASTRUCTURE ONLY - plain-C CPU restatement of the reference-real pixel ops.
 * Only tests/ (and __graft_entry__.build(), which compiles it) touch this file; the product
 * never links or calls it.  Pinned against tests/golden/image_losses_*.npz, adam_beta05.npz and
 * prep_input.npz, which were produced by the reference itself (oracle/make_golden.py).
 *
 * Each function cites the reference lines it restates (paths under /root/reference).
 * Accumulation is in double so the oracle is tighter than either fp32 implementation.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static double sgn(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); }

/* nn.CrossEntropyLoss(reduction='mean') on (b,C,H,W) logits, (b,H,W) int64 targets
 * - reference src/trainer.py:124 (construction), :250 (call, x10 applied by the caller) */
double oracle_ce_nchw(const float* logits, const int64_t* target, float* dlogits, int b, int C, int64_t hw) {
    const int64_t npix = (int64_t)b * hw;
    double total = 0.0;
    int64_t cnt = 0;
    for (int64_t i = 0; i < npix; ++i) cnt += target[i] != -100;
    const double inv = cnt > 0 ? 1.0 / (double)cnt : 0.0;
    for (int64_t i = 0; i < npix; ++i) {
        const int64_t n = i / hw, p = i % hw;
        const float* row = logits + n * C * hw + p;
        double mx = row[0];
        for (int c = 1; c < C; ++c) if (row[c * hw] > mx) mx = row[c * hw];
        double se = 0.0;
        for (int c = 0; c < C; ++c) se += exp((double)row[c * hw] - mx);
        const int64_t t = target[i];
        if (t != -100) total += log(se) + mx - (double)row[t * hw];
        if (dlogits) {
            float* drow = dlogits + n * C * hw + p;
            for (int c = 0; c < C; ++c) {
                const double sm = exp((double)row[c * hw] - mx) / se;
                drow[c * hw] = (float)(t == -100 ? 0.0 : inv * (sm - (c == t ? 1.0 : 0.0)));
            }
        }
    }
    return total * inv;
}

/* nn.L1Loss() - reference src/trainer.py:130 (construction), :248 (call) */
double oracle_l1_mean(const float* a, const float* b, float* da, int64_t n) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const double d = (double)a[i] - (double)b[i];
        s += fabs(d);
        if (da) da[i] = (float)(sgn(d) / (double)n);
    }
    return s / (double)n;
}

/* GradientLoss.forward - reference src/loss.py:20-25
 *   xloss = sum | |a[h+1]-a[h]| - |b[h+1]-b[h]| |, yloss likewise along w, / numel */
double oracle_gradient_loss(const float* a, const float* b, float* da, int planes, int H, int W) {
    const int64_t n = (int64_t)planes * H * W;
    double s = 0.0;
    double* g = da ? (double*)calloc((size_t)n, sizeof(double)) : NULL;
    for (int p = 0; p < planes; ++p)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                const int64_t i = ((int64_t)p * H + h) * W + w;
                if (h + 1 < H) {
                    const double d_a = (double)a[i + W] - a[i], d_b = (double)b[i + W] - b[i];
                    const double e = fabs(d_a) - fabs(d_b);
                    s += fabs(e);
                    if (g) { g[i + W] += sgn(e) * sgn(d_a); g[i] -= sgn(e) * sgn(d_a); }
                }
                if (w + 1 < W) {
                    const double d_a = (double)a[i + 1] - a[i], d_b = (double)b[i + 1] - b[i];
                    const double e = fabs(d_a) - fabs(d_b);
                    s += fabs(e);
                    if (g) { g[i + 1] += sgn(e) * sgn(d_a); g[i] -= sgn(e) * sgn(d_a); }
                }
            }
    if (g) {
        for (int64_t i = 0; i < n; ++i) da[i] = (float)(g[i] / (double)n);
        free(g);
    }
    return s / (double)n;
}

/* SsimLoss.SSIM / forward - reference src/loss.py:68-91
 *   per channel: 3x3 avg_pool (stride 1, no padding) of x, y, x^2, y^2, xy; C1 = 0.01^2, C2 = 0.03^2;
 *   clamp((1 - SSIM)/2, 0, 1).mean(); the per-channel means are SUMMED (loss.py:89-91) */
double oracle_ssim_loss(const float* x, const float* y, float* dx, int b, int C, int H, int W) {
    const double C1 = 0.01 * 0.01, C2 = 0.03 * 0.03;
    const int64_t hw = (int64_t)H * W, n = (int64_t)b * C * hw;
    const double wnd = 1.0 / ((double)b * (H - 2) * (W - 2));
    double total = 0.0;
    double* g = dx ? (double*)calloc((size_t)n, sizeof(double)) : NULL;
    for (int64_t pl = 0; pl < (int64_t)b * C; ++pl) {
        const float* xp = x + pl * hw;
        const float* yp = y + pl * hw;
        for (int h = 1; h + 1 < H; ++h)
            for (int w = 1; w + 1 < W; ++w) {
                double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
                for (int dh = -1; dh <= 1; ++dh)
                    for (int dw = -1; dw <= 1; ++dw) {
                        const double xv = xp[(h + dh) * W + w + dw], yv = yp[(h + dh) * W + w + dw];
                        sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv;
                    }
                const double mux = sx / 9, muy = sy / 9;
                const double sgx = sxx / 9 - mux * mux, sgy = syy / 9 - muy * muy, sgxy = sxy / 9 - mux * muy;
                const double A1 = 2 * mux * muy + C1, A2 = 2 * sgxy + C2;
                const double B1 = mux * mux + muy * muy + C1, B2 = sgx + sgy + C2;
                const double S = A1 * A2 / (B1 * B2);
                double v = (1 - S) / 2;
                const int pass = v >= 0 && v <= 1;
                if (v < 0) v = 0;
                if (v > 1) v = 1;
                total += v * wnd;
                if (g && pass) {
                    const double inv = 1.0 / (B1 * B2);
                    const double dmu = (2 * muy * (A2 - A1) - 2 * mux * S * (B2 - B1)) * inv;
                    const double dexx = -S / B2, dexy = 2 * A1 * inv;
                    for (int dh = -1; dh <= 1; ++dh)
                        for (int dw = -1; dw <= 1; ++dw) {
                            const int64_t q = (int64_t)(h + dh) * W + w + dw;
                            g[pl * hw + q] += -0.5 * wnd * (dmu / 9 + dexx * 2 * xp[q] / 9 + dexy * yp[q] / 9);
                        }
                }
            }
    }
    if (g) {
        for (int64_t i = 0; i < n; ++i) dx[i] = (float)g[i];
        free(g);
    }
    return total;
}

/* torch.optim.Adam(lr, betas=(beta1, 0.999)), eps 1e-8, no weight decay, no amsgrad
 * - reference src/trainer.py:83 (construction), :258 (step); src/main.py:139-141 (lr, beta1) */
void oracle_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step, double lr, double beta1,
                      double beta2, double eps) {
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    for (int64_t i = 0; i < n; ++i) {
        const double mi = beta1 * m[i] + (1 - beta1) * g[i];
        const double vi = beta2 * v[i] + (1 - beta2) * (double)g[i] * g[i];
        m[i] = (float)mi;
        v[i] = (float)vi;
        p[i] = (float)(p[i] - (lr / bc1) * mi / (sqrt(vi) / sqrt(bc2) + eps));
    }
}

/* input preparation - reference src/trainer.py:193-206
 *   frames -> (f - mean) / std (:193-195); x = cat[e1, seg1, f1, f2, seg2, e2] (:197);
 *   flip: x, frame3 on W (:202-205), seg3 on its last dim (:206) */
void oracle_prep_input(const float* e1, const float* seg1, const float* f1, const float* f2, const float* seg2,
                       const float* e2, const float* f3, const int64_t* seg3, float* x10, float* f3o,
                       int64_t* seg3o, int b, int H, int W, int flip) {
    static const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const int64_t hw = (int64_t)H * W;
    for (int n = 0; n < b; ++n)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                const int ws = flip ? W - 1 - w : w;
                const int64_t s1 = n * hw + (int64_t)h * W + ws, r = (int64_t)h * W + w;
                float* xo = x10 + (int64_t)n * 10 * hw + r;
                xo[0] = e1[s1];
                xo[hw] = seg1[s1];
                for (int c = 0; c < 3; ++c) {
                    const int64_t s3 = ((int64_t)n * 3 + c) * hw + (int64_t)h * W + ws;
                    xo[(2 + c) * hw] = (f1[s3] - mean[c]) / stdv[c];
                    xo[(5 + c) * hw] = (f2[s3] - mean[c]) / stdv[c];
                    f3o[((int64_t)n * 3 + c) * hw + r] = (f3[s3] - mean[c]) / stdv[c];
                }
                xo[8 * hw] = seg2[s1];
                xo[9 * hw] = e2[s1];
                seg3o[n * hw + r] = seg3[s1];
            }
}
