/* ORACLE — test infrastructure only.  See uct_oracle.h for provenance and pinning. */
#include "uct_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- glibc 2.35 stdlib/random_r.c, TYPE_3 (degree 31, separation 3) ---- */
void orc_srand(orc_rand_t *s, uint32_t seed) {
    if (seed == 0) seed = 1;
    int32_t word = (int32_t)seed;
    s->r[0] = word;
    for (int i = 1; i < 31; ++i) {
        long hi = word / 127773, lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int32_t)w;
        s->r[i] = word;
    }
    s->f = 3;
    s->b = 0;
    for (int i = 0; i < 310; ++i) (void)orc_rand(s);
}
int32_t orc_rand(orc_rand_t *s) {
    uint32_t val = (uint32_t)s->r[s->f] + (uint32_t)s->r[s->b];
    s->r[s->f] = (int32_t)val;
    s->f += 1;
    if (s->f >= 31) { s->f = 0; s->b += 1; }
    else { s->b += 1; if (s->b >= 31) s->b = 0; }
    return (int32_t)(val >> 1);
}

/* special.h:26-33 */
double orc_norm_quantile(double t) {
    double log2_ = log(2), log22 = log(22), log41 = log(41);
    double alpha = 1 - 1 / t;
    return 10 * log(1 - log(-log(alpha) / log2_) / log22) / log41;
}

/* core.h:111-144 */
int orc_get_unique_child_obs(int index, const int32_t *child, const float *score, const int32_t *n_to_o,
                             int32_t *c_nodes, int32_t *c_obs) {
    int n = 0;
    for (int i = 0; i < ORC_NACT; ++i) {
        int c = child[(size_t)index * ORC_NACT + i];
        if (c == 0) continue;
        int o = n_to_o[c];
        int found = -1;
        for (int j = 0; j < n; ++j)
            if (c_obs[j] == o) { found = j; break; }
        if (found < 0) {
            c_nodes[n] = c;
            c_obs[n] = o;
            n += 1;
        } else if (score[c] > score[c_nodes[found]]) {
            c_nodes[found] = c;
        }
    }
    return n;
}

/* core.h:65-77 */
int orc_check_low(const int32_t *c_obs, int n, const int32_t *visit, int low, orc_rand_t *rng) {
    int32_t lows[ORC_NACT];
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (visit[c_obs[i]] < low) lows[m++] = c_obs[i];
    if (m == 0) return 0;
    return lows[orc_rand(rng) % m];
}

/* core.h:83-105 : float arithmetic, one rounding per operation (no FMA) */
int orc_policy_clt(const int32_t *nodes, const int32_t *visit, const float *value, const float *variance, int n) {
    int total = 0;
    for (int i = 0; i < n; ++i) total += visit[i];
    int max_idx = 0;
    float max_q = 0;
    float bound_coeff = (float)orc_norm_quantile((double)total);
    for (int i = 0; i < n; ++i) {
        float ratio = variance[i] / (float)visit[i];
        float root = sqrtf(ratio);
        float prod = bound_coeff * root;
        float q = value[i] + prod;
        if (i == 0) max_q = q;
        else if (q > max_q) { max_q = q; max_idx = i; }
    }
    return nodes[max_idx];
}

/* core.h:167-224 */
int orc_select_trace_obs(int index, const int32_t *child, const int32_t *visit, const float *value,
                         const float *variance, const float *score, const int32_t *n_to_o, int low,
                         orc_rand_t *rng, int32_t *trace, int cap) {
    int len = 0;
    int32_t c_nodes[ORC_NACT], c_obs[ORC_NACT], _visit[ORC_NACT];
    float _value[ORC_NACT], _variance[ORC_NACT];
    for (;;) {
        if (len >= cap) return -1;
        trace[len++] = index;
        int n = orc_get_unique_child_obs(index, child, score, n_to_o, c_nodes, c_obs);
        if (n == 0) break;
        int o = orc_check_low(c_obs, n, visit, low, rng);
        if (o == 0) {
            for (int i = 0; i < n; ++i) {
                _visit[i] = visit[c_obs[i]];
                float t = value[c_obs[i]] + score[c_nodes[i]];
                _value[i] = t - score[index];
                _variance[i] = variance[c_obs[i]];
            }
            index = orc_policy_clt(c_nodes, _visit, _value, _variance, n);
        } else {
            for (int i = 0; i < n; ++i)
                if (c_obs[i] == o) { index = c_nodes[i]; break; }
        }
    }
    return len;
}

/* core.h:226-260 */
void orc_backup_trace_obs(const int32_t *trace, int len, int32_t *visit, float *value, float *variance,
                          const int32_t *n_to_o, const float *score, double _value, double _variance, double gamma) {
    for (int i = len - 1; i >= 0; --i) {
        int idx = trace[i];
        _value -= (double)score[idx];
        int o = n_to_o[idx];
        if (visit[o] == 0) {
            value[o] = (float)_value;
            variance[o] = (float)_variance;
        } else {
            double delta = _value - (double)value[o];
            value[o] = (float)((double)value[o] + delta / (double)(visit[o] + 1));
            double delta2 = _value - (double)value[o];
            double prod = delta * delta2;
            variance[o] = (float)((double)variance[o] + (prod - (double)variance[o]) / (double)(visit[o] + 1));
        }
        visit[o] += 1;
        double t = gamma * _value;
        _value = t + (double)score[idx];
    }
}

/* core.h:262-301 */
void orc_backup_trace_mixture_obs(const int32_t *trace, int len, int32_t *visit, float *value, float *variance,
                                  const int32_t *n_to_o, const float *score, double _value, double _variance,
                                  double gamma) {
    for (int i = len - 1; i >= 0; --i) {
        int idx = trace[i];
        _value -= (double)score[idx];
        int o = n_to_o[idx];
        visit[o] += 1;
        double a = _value * _value;
        double b = (double)(value[o] * value[o]); /* float*float product, rounded to float first */
        double v_sq_diff = a - b;
        double v_tmp = (double)value[o];
        double delta = (_value - (double)value[o]) / (double)visit[o];
        value[o] = (float)((double)value[o] + delta);
        double var_diff = _variance - (double)variance[o];
        double t1 = (var_diff + v_sq_diff) / (double)visit[o];
        double t2 = delta * (v_tmp + (double)value[o]);
        variance[o] = (float)(((double)variance[o] + t1) - t2);
        double t = gamma * _value;
        _value = t + (double)score[idx];
        _variance *= (gamma * gamma);
    }
}

/* core.h:303-381 */
void orc_backup_trace_obs_LP(const int32_t *trace, int len, int32_t *visit, float *value, float *variance,
                             const int32_t *n_to_o, const float *score, const uint8_t *end, const int32_t *_child,
                             const int32_t *_obs, int k, const float *_value, const float *_variance, double gamma,
                             int mixture, int averaged) {
    void (*backup)(const int32_t *, int, int32_t *, float *, float *, const int32_t *, const float *, double, double,
                   double) = mixture ? orc_backup_trace_mixture_obs : orc_backup_trace_obs;
    if (k > 0) {
        double v_tmp = 0, var_tmp = 0;
        for (int i = 0; i < k; ++i) {
            int c = _child[i], o = _obs[i];
            if (visit[o] == 0) {
                visit[o] += 1;
                if (end[c]) { value[o] = 0; variance[o] = 0; }
                else { value[o] = _value[i]; variance[o] = _variance[i]; }
            }
            if (averaged) {
                double gv = gamma * (double)value[o];
                v_tmp += (double)score[c] + gv;
                var_tmp += (double)variance[o];
            } else {
                double gs = gamma * (double)score[c];
                double gg = gamma * gamma;
                backup(trace, len, visit, value, variance, n_to_o, score, (double)value[o] + gs,
                       gg * (double)variance[o], gamma);
            }
        }
        if (averaged) {
            v_tmp /= (double)k;
            var_tmp *= (gamma * gamma / (double)k);
            backup(trace, len, visit, value, variance, n_to_o, score, v_tmp, var_tmp, gamma);
        }
    } else {
        backup(trace, len, visit, value, variance, n_to_o, score, (double)score[trace[len - 1]], 0, gamma);
    }
}

/* agent.cpp:496-513 : the carried value is a float */
void orc_backup_obs_single_cppagent(const int32_t *trace, int len, int32_t *visit, float *value, float *variance,
                                    const int32_t *n_to_o, const float *score, float _val, float _var, double gamma) {
    for (int i = len - 1; i >= 0; --i) {
        int c = trace[i];
        int o = n_to_o[c];
        _val -= score[c];
        if (visit[o] == 0) {
            value[o] = _val;
            variance[o] = _var;
        } else {
            double delta = (double)(_val - value[o]);
            value[o] = (float)((double)value[o] + delta / (double)(visit[o] + 1));
            double delta2 = (double)(_val - value[o]);
            double prod = delta * delta2;
            variance[o] = (float)((double)variance[o] + (prod - (double)variance[o]) / (double)(visit[o] + 1));
        }
        visit[o] += 1;
        double t = gamma * (double)_val;
        _val = (float)(t + (double)score[c]);
    }
}

/* agent.cpp:517-566 with mixture=false, averaged=true (the only combination MCTSAgent::mcts uses) */
void orc_backup_obs_cppagent(const int32_t *trace, int len, int32_t *visit, float *value, float *variance,
                             const int32_t *n_to_o, const float *score, const uint8_t *end_obs,
                             const int32_t *_child, const int32_t *_obs, int k, const float *_value,
                             const float *_variance, double gamma, float leaf_score) {
    if (k == 0) {
        orc_backup_obs_single_cppagent(trace, len, visit, value, variance, n_to_o, score, leaf_score, 0, gamma);
        return;
    }
    double v = 0, var = 0;
    for (int i = 0; i < k; ++i) {
        int c = _child[i], o = _obs[i];
        if (visit[o] == 0) {
            visit[o] += 1;
            if (end_obs[o]) { value[o] = 0; variance[o] = 0; }
            else { value[o] = _value[i]; variance[o] = _variance[i]; }
        }
        double gv = gamma * (double)value[o];
        v += (double)score[c] + gv;
        var += (double)variance[o];
    }
    v /= (double)k;
    var /= (double)k;
    orc_backup_obs_single_cppagent(trace, len, visit, value, variance, n_to_o, score, (float)v, (float)var, gamma);
}

/* core.h:32-50 */
int orc_get_all_childs(int index, const int32_t *child, int n_nodes, uint8_t *mark) {
    int32_t *queue = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_nodes);
    memset(mark, 0, (size_t)n_nodes);
    int head = 0, tail = 0, count = 0;
    queue[tail++] = index;
    mark[index] = 1;
    count = 1;
    while (head < tail) {
        int n = queue[head++];
        for (int a = 0; a < ORC_NACT; ++a) {
            int c = child[(size_t)n * ORC_NACT + a];
            if (!mark[c]) { mark[c] = 1; count += 1; queue[tail++] = c; }
        }
    }
    free(queue);
    return count;
}

/* The same set as orc_get_all_childs (core.h:32-50) by the SCHEDULE of the device's collector (tree.hip, gc_sweep_mark):
 * no queue - two bitmaps, "marked" and "pending" (marked, children not looked at yet), and sweeps over the pending bitmap
 * in DESCENDING index order (free indices are popped highest first, agents/agent.py:211-212 + deque.pop(): a node's
 * children mostly lie below it, so one sweep carries a whole chain).  A sweep goes segment by segment (seg_nodes indices);
 * a segment is re-read until it holds nothing pending; the nodes of one reading are processed TOGETHER (their child rows
 * are loaded before any of their marks is set: the device's concurrency), so a mark made by a reading is seen by the
 * next reading, never by its own.  A child above the segment under work is left for the next sweep, which starts at the
 * highest such segment.  Node 0 is marked when a processed row holds a zero (the reference follows the zero entries of a
 * row like any child, core.h:41-45) and never followed itself (its own row is all zero).  stats (may be NULL): [0] sweeps, [1] segment readings that found something,
 * [2] child rows loaded (= nodes processed), [3] segment readings in all.  Returns the number of marked nodes. */
int orc_sweep_marks(int index, const int32_t *child, int n_nodes, uint8_t *mark, int seg_nodes, long *stats) {
    uint8_t *pend = (uint8_t *)calloc((size_t)n_nodes, 1);
    int32_t *batch = (int32_t *)malloc(sizeof(int32_t) * (size_t)seg_nodes);
    memset(mark, 0, (size_t)n_nodes);
    const int n_seg = (n_nodes + seg_nodes - 1) / seg_nodes;
    long sweeps = 0, readings_hit = 0, rows = 0, readings = 0;
    int count = 1;
    mark[index] = 1;
    if (index != 0) pend[index] = 1;
    int start = index / seg_nodes;                       /* highest segment that may hold something pending */
    while (start >= 0) {
        int next_start = -1;
        sweeps += 1;
        for (int s = start; s >= 0; --s) {
            const int lo = s * seg_nodes, hi = lo + seg_nodes < n_nodes ? lo + seg_nodes : n_nodes;
            for (;;) {
                int nb = 0;
                readings += 1;
                for (int i = hi - 1; i >= lo; --i)
                    if (pend[i]) { batch[nb++] = i; pend[i] = 0; }
                if (nb == 0) break;
                readings_hit += 1;
                rows += nb;
                for (int b = 0; b < nb; ++b)
                    for (int a = 0; a < ORC_NACT; ++a) {
                        const int c = child[(size_t)batch[b] * ORC_NACT + a];
                        if (mark[c]) continue;
                        mark[c] = 1; count += 1;
                        if (c == 0) continue;
                        pend[c] = 1;
                        const int cs = c / seg_nodes;
                        if (cs > s && cs > next_start) next_start = cs;
                    }
            }
        }
        start = next_start;
    }
    if (stats) { stats[0] = sweeps; stats[1] = readings_hit; stats[2] = rows; stats[3] = readings; }
    (void)n_seg;
    free(pend); free(batch);
    return count;
}
