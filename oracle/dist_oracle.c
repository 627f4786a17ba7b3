/*
 * ORACLE — test infrastructure only.  CPU restatement of the reference's distribution helpers
 * (agents/cppmodule/core.h:387-449; defined there, not exported by core.cpp:20-26, pinned through
 * oracle/ref_dist_shim.cpp -> tests/golden/ref_dist.npz).
 *
 * transform_distribution (core.h:387-409): every source bin b is an interval of width `scale` bins starting at
 * lb = max(b*scale + shift/delta, 0); its mass is split between the two destination bins it overlaps.  The reference
 * writes result[b_ub] even when b_ub == bins (one float past its vector: heap corruption whenever that mass is not 0);
 * here - and on the device - mass that falls beyond the last bin is dropped, which is what the returned array shows in
 * every case the reference survives.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

void orc_transform_distribution(const float *dist, int bins, double vmin, double vmax, double shift, double scale,
                                float *result) {
    for (int b = 0; b < bins; ++b) result[b] = 0.0f;
    double delta = (vmax - vmin) / bins;
    double bin_shift = shift / delta;
    for (int b = 0; b < bins; ++b) {
        double lb = fmax(b * scale + bin_shift, 0.);
        int b_lb = (int)floor(lb);
        double ub = fmin(lb + scale, (double)bins);
        int b_ub = (int)floor(ub);
        double frac = b_ub - lb;
        if (b_lb < bins) result[b_lb] = (float)((double)result[b_lb] + (double)dist[b] * frac);   /* float += double */
        if (b_ub < bins) result[b_ub] = (float)((double)result[b_ub] + (double)dist[b] * (1 - frac));
    }
}

double orc_mean_dist(const float *dist, int bins, double vmin, double vmax) {
    double delta = (vmax - vmin) / bins, mean = 0, center = vmin + 0.5 * delta;
    for (int b = 0; b < bins; ++b) {
        mean += center * dist[b];
        center += delta;
    }
    return mean;
}

void orc_mean_variance_dist(const float *dist, int bins, double vmin, double vmax, double *out /* mean, var */) {
    double delta = (vmax - vmin) / bins, mean = 0, m2 = 0, center = vmin + 0.5 * delta;
    for (int b = 0; b < bins; ++b) {
        double tmp = center * dist[b];
        mean += tmp;
        m2 += center * tmp;
        center += delta;
    }
    out[0] = mean;
    out[1] = m2 - mean * mean;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * agents/core_distributional.py (numba kernels of the distributional agent the reference never finished):
 * shift_distribution :12-37, policy_dist :66-79, backup_trace_distributional :108-124.  They are `fastmath` numba code, so
 * there is no bit pattern to reproduce: restated with numba's typing rules (float32 arrays, float64 scalars, a float32
 * array element op a float64 scalar is done in double and rounded on the store) and compared with a pure-Python run of
 * the reference functions (numba shimmed away; tests/golden/ref_distpy.npz) within a float tolerance.
 * mean_dist / mean_variance (:40-63) are core.h's helpers with bin centres measured from 0: orc_mean_variance_dist(dist,
 * bins, 0, vmax - vmin).
 * ------------------------------------------------------------------------------------------------------------------ */
double orc_norm_quantile(double t); /* uct_oracle.c (special.h:26-33 == agents/special.py:56-66) */

static int py_index(int i, int n) { return i < 0 ? i + n : i; } /* Python / numba wraparound indexing */

void orc_distpy_shift(const float *dist, int bins, double x, double vmin, double vmax, float *result) {
    double delta = (vmax - vmin) / bins;
    for (int b = 0; b < bins; ++b) result[b] = 0.0f;
    double bin_shift = x / delta;
    double fraction = bin_shift - floor(bin_shift);
    for (int b = 0; b < bins; ++b) {
        int b_lb = (int)(b + bin_shift); /* int(): truncation toward zero */
        if (b_lb >= bins) b_lb = bins - 1;
        int b_ub = (b_lb + 1 >= bins) ? bins - 1 : b_lb + 1;
        int lo = py_index(b_lb, bins), hi = py_index(b_ub, bins);
        if (lo >= 0) result[lo] = (float)((double)result[lo] + (double)dist[b] * (1 - fraction));
        if (hi >= 0) result[hi] = (float)((double)result[hi] + (double)dist[b] * fraction);
    }
}

int orc_distpy_policy(const int32_t *child_nodes, int nc, const float *node_stats /* [n][5] */, double curr_reward) {
    const double eps = 1e-3; /* core_distributional.py:8 */
    double n = 0;
    float s0[7], s1[7];
    for (int i = 0; i < nc; ++i) {
        const float *ns = node_stats + (size_t)child_nodes[i] * 5;
        n += ns[0];
        float t = ns[1] + ns[2];
        s0[i] = (float)((double)t - curr_reward);
        s1[i] = (float)((double)ns[3] / ((double)ns[0] + eps));
    }
    double coeff = orc_norm_quantile(n);
    int best = 0;
    double best_q = 0;
    for (int i = 0; i < nc; ++i) {
        double q = (double)s0[i] + coeff * (double)sqrtf(s1[i]);
        if (i == 0) { best_q = q; continue; }
        if (best_q != best_q) break;                      /* np.argmax: the first NaN wins */
        if (q != q || q > best_q) { best_q = q; best = i; }
    }
    return child_nodes[best];
}

void orc_distpy_backup(const int32_t *trace, int len, float *node_stats /* [n][5] */, float *node_dist /* [n][bins] */,
                       int bins, double r, const float *dist, double vmin, double vmax, float *scratch /* [bins] */) {
    double delta = (vmax - vmin) / bins, mean = 0;
    for (int b = 0; b < bins; ++b) mean += (double)dist[b] * ((b + 0.5) * delta);   /* mean_dist :40-46 */
    for (int t = 0; t < len; ++t) {
        int idx = trace[t];
        float *ns = node_stats + (size_t)idx * 5, *nd = node_dist + (size_t)idx * bins;
        double _r = r - (double)ns[2];
        orc_distpy_shift(dist, bins, _r, vmin, vmax, scratch);
        for (int b = 0; b < bins; ++b) {
            float m = nd[b] * ns[0];
            float u = m + scratch[b];
            nd[b] = (float)((double)u / ((double)ns[0] + 1.0));
        }
        double x = mean + _r;
        ns[0] = ns[0] + 1.0f;
        double d1 = x - (double)ns[1];
        ns[1] = (float)((double)ns[1] + d1 / (double)ns[0]);
        double d2 = x - (double)ns[1];
        ns[4] = (float)((double)ns[4] + d1 * d2);
        if (ns[0] > 1.0f) ns[3] = (float)((double)ns[4] / ((double)ns[0] - 1.0));
    }
}
