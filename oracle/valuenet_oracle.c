/*
 * ORACLE — test infrastructure only (see oracle/tetris_engine.h).
 *
 * CPU restatement of the reference's value network forward pass, model/model_vv.py:13-52
 * (Net: 3x conv3x3(32)+ReLU, flatten, fc 1792->256 + ReLU, fc 256->2, sigmoid, affine) and
 * Model_VV.inference model/model_vv.py:210-217.
 *
 * Every output element is ONE fp32 fused-multiply-add chain: acc = bias; for k ascending
 * (k = ci*9 + ky*3 + kx for the convolutions, k = flat input index for the linears):
 * acc = fmaf(x_k, w_k, acc).  That is exactly what gfx950's v_mfma_f32_32x32x2_f32 computes
 * (a k-ordered fmaf chain, one rounding per product), so the HIP kernel and this file agree
 * bit for bit; both stay within 1e-4 of torch's fp32 result (tests/test_valuenet*.py, which pin
 * this file against the reference's own Net imported from /root/reference -> tests/golden/).
 * The sigmoid uses orc_exp (explicit fma polynomial, identical source on both sides).
 *
 * Parameter blob (floats, PyTorch state_dict order and layouts):
 *   conv1.w[32][1][3][3] conv1.b[32] conv2.w[32][32][3][3] conv2.b[32] conv3.w[32][32][3][3]
 *   conv3.b[32] fc1.w[256][1792] fc1.b[256] fc_out.w[2][256] fc_out.b[2] out_ubound[2] out_lbound[2]
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define VN_PARAMS 478342

double orc_exp(double x) {
    /* exp(x) = 2^n * exp(r), r = x - n*ln2, |r| <= ln2/2 ; degree-13 Taylor in Horner form (fma) */
    if (x > 700.0) x = 700.0;
    if (x < -700.0) x = -700.0;
    const double inv_ln2 = 1.4426950408889634074, ln2_hi = 6.93147180369123816490e-01,
                 ln2_lo = 1.90821492927058770002e-10;
    double n = nearbyint(x * inv_ln2);
    double r = fma(-n, ln2_hi, x);
    r = fma(-n, ln2_lo, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int64_t bits;
    memcpy(&bits, &p, 8);
    bits += ((int64_t)n) << 52; /* |n| <= 1010 and p in [0.7, 1.5]: never leaves the normal range here */
    memcpy(&p, &bits, 8);
    return p;
}

static void conv3x3_relu(const float *in, int cin, int h, int w, const float *wt, const float *b, float *out) {
    int oh = h - 2, ow = w - 2;
    for (int co = 0; co < 32; ++co)
        for (int y = 0; y < oh; ++y)
            for (int x = 0; x < ow; ++x) {
                float acc = b[co];
                for (int ci = 0; ci < cin; ++ci)
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx)
                            acc = fmaf(in[(ci * h + y + ky) * w + x + kx], wt[((co * cin + ci) * 3 + ky) * 3 + kx], acc);
                out[(co * oh + y) * ow + x] = acc > 0.0f ? acc : 0.0f;
            }
}

/* states: k x 200 int8 in {-1,0,1}; v/var: k floats each */
void orc_valuenet_forward(const float *P, const int8_t *states, int k, float *v, float *var) {
    const float *c1w = P, *c1b = c1w + 288, *c2w = c1b + 32, *c2b = c2w + 9216, *c3w = c2b + 32, *c3b = c3w + 9216,
                *f1w = c3b + 32, *f1b = f1w + 458752, *fow = f1b + 256, *fob = fow + 512, *ub = fob + 2, *lb = ub + 2;
    float x0[200], a1[32 * 18 * 8], a2[32 * 16 * 6], a3[1792], h[256];   /* on the stack: agents run in parallel threads in the tests */
    for (int s = 0; s < k; ++s) {
        for (int i = 0; i < 200; ++i) x0[i] = (float)states[200 * s + i];
        conv3x3_relu(x0, 1, 20, 10, c1w, c1b, a1);
        conv3x3_relu(a1, 32, 18, 8, c2w, c2b, a2);
        conv3x3_relu(a2, 32, 16, 6, c3w, c3b, a3);
        for (int j = 0; j < 256; ++j) {
            float acc = f1b[j];
            for (int i = 0; i < 1792; ++i) acc = fmaf(a3[i], f1w[j * 1792 + i], acc);
            h[j] = acc > 0.0f ? acc : 0.0f;
        }
        float out[2];
        for (int j = 0; j < 2; ++j) {
            float acc = fob[j];
            for (int i = 0; i < 256; ++i) acc = fmaf(h[i], fow[j * 256 + i], acc);
            double e = orc_exp(-(double)acc);
            float sg = (float)(1.0 / (1.0 + e));
            float t = sg * ub[j];
            out[j] = t + lb[j];
        }
        v[s] = out[0];
        var[s] = out[1];
    }
}

/* evaluator callback with the orc_eval_fn signature; ctx = parameter blob */
void orc_valuenet_eval(void *ctx, const int8_t *states, int k, float *v, float *var) {
    orc_valuenet_forward((const float *)ctx, states, k, v, var);
}
