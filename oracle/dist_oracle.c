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
