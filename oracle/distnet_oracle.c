/*
 * ORACLE — test infrastructure only (see oracle/tetris_engine.h).
 *
 * CPU restatement of the reference's distributional value head, model/model_distributional.py:18-57 (`Net`: conv 4x4 1->32,
 * LeakyReLU, conv 4x4 32->32, LeakyReLU, flatten, fc 2048->128, LeakyReLU, fc 128->atoms; forward = softmax over the atoms,
 * :44-48) and Model_Dist.inference (:100-107), on the reference's hard-wired 22 x 10 input (:27) = the 20 visible rows of a
 * state under two empty ones (how tetris_mcts_amd/agents/DistValueSim.py feeds it).
 *
 * Same contract as valuenet_oracle.c: every pre-activation is ONE fp32 fused-multiply-add chain, acc = bias; for k ascending
 * (k = ci*16 + ky*4 + kx for the convolutions, the flat input index for the linear layers): acc = fmaf(x_k, w_k, acc) - what
 * gfx950's fp32 MFMA instructions compute per output, so tetris_mcts_amd/csrc/distnet.hip agrees bit for bit.  LeakyReLU:
 * x > 0 ? x : x * 0.01f (torch's CPU kernel, negative_slope cast to float).  Softmax: m = max, e_b = orc_exp((double)x_b -
 * (double)m), sum over b ascending in double, p_b = (float)(e_b / sum).  Pinned on the reference's own Net run on CPU
 * (tests/golden/ref_distnet.npz, 1e-6 relative: tests/test_oracle_dist.py).
 *
 * Parameter blob (floats, PyTorch state_dict order and layouts):
 *   seq.conv1.w[32][1][4][4] seq.conv1.b[32] seq.conv2.w[32][32][4][4] seq.conv2.b[32] seq.fc1.w[128][2048] seq.fc1.b[128]
 *   seq.fc_v.w[atoms][128] seq.fc_v.b[atoms]
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

double orc_exp(double x);   /* valuenet_oracle.c */

static float leaky(float v) { return v > 0.0f ? v : v * 0.01f; }

static void conv4x4_leaky(const float *in, int cin, int h, int w, const float *wt, const float *b, float *out) {
    int oh = h - 3, ow = w - 3;
    for (int co = 0; co < 32; ++co)
        for (int y = 0; y < oh; ++y)
            for (int x = 0; x < ow; ++x) {
                float acc = b[co];
                for (int ci = 0; ci < cin; ++ci)
                    for (int ky = 0; ky < 4; ++ky)
                        for (int kx = 0; kx < 4; ++kx)
                            acc = fmaf(in[(ci * h + y + ky) * w + x + kx], wt[((co * cin + ci) * 4 + ky) * 4 + kx], acc);
                out[(co * oh + y) * ow + x] = leaky(acc);
            }
}

/* states: k x 200 int8 in {-1,0,1} (20 x 10); dist: k x atoms floats */
void orc_distnet_forward(const float *P, const int8_t *states, int k, int atoms, float *dist) {
    const float *c1w = P, *c1b = c1w + 512, *c2w = c1b + 32, *c2b = c2w + 16384, *f1w = c2b + 32, *f1b = f1w + 262144,
                *fvw = f1b + 128, *fvb = fvw + (size_t)atoms * 128;
    float x0[220], a1[32 * 19 * 7], a2[2048], h[128], lg[64];
    double e[64];
    for (int s = 0; s < k; ++s) {
        for (int i = 0; i < 20; ++i) x0[i] = 0.0f;
        for (int i = 0; i < 200; ++i) x0[20 + i] = (float)states[200 * s + i];
        conv4x4_leaky(x0, 1, 22, 10, c1w, c1b, a1);
        conv4x4_leaky(a1, 32, 19, 7, c2w, c2b, a2);
        for (int j = 0; j < 128; ++j) {
            float acc = f1b[j];
            for (int i = 0; i < 2048; ++i) acc = fmaf(a2[i], f1w[(size_t)j * 2048 + i], acc);
            h[j] = leaky(acc);
        }
        float m = -INFINITY;
        for (int o = 0; o < atoms; ++o) {
            float acc = fvb[o];
            for (int i = 0; i < 128; ++i) acc = fmaf(h[i], fvw[(size_t)o * 128 + i], acc);
            lg[o] = acc;
            if (acc > m) m = acc;
        }
        double sum = 0.0;
        for (int o = 0; o < atoms; ++o) { e[o] = orc_exp((double)lg[o] - (double)m); sum = sum + e[o]; }
        for (int o = 0; o < atoms; ++o) dist[(size_t)s * atoms + o] = (float)(e[o] / sum);
    }
}

/* evaluator callback with the orc_dist_eval_fn signature (agent_oracle.c kind 6); ctx = parameter blob */
void orc_distnet_eval(void *ctx, const int8_t *states, int k, int bins, float *dist) {
    orc_distnet_forward((const float *)ctx, states, k, bins, dist);
}
