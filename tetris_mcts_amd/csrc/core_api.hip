// agents.cppmodule.core (agents/cppmodule/core.cpp:20-26) in batched form: B independent trees stored in the
// reference's own array layout (agents/agent.py:58-88), tree b at offset b*n_nodes of every array.
// One lane per tree: this is the compatibility surface (a drop-in for callers that own their numpy-style
// arrays); the performance path is the fused one-wave-per-game engine in tree.hip.
//   select_trace_obs core.h:167-224 | backup_trace_obs core.h:226-260 | backup_trace_obs_LP core.h:303-381
//   get_unique_child_obs core.h:111-144 | get_all_childs core.h:32-50
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/tetris_mcts_hip.h"

namespace tmcts_core {

__device__ inline float nq(const float* table, int size, int n) {
    if (n >= 0 && n < size) return table[n];
    double t = (double)n, alpha = 1 - 1 / t;
    return (float)(10 * log(1 - log(-log(alpha) / log(2.0)) / log(22.0)) / log(41.0));
}

// rng: 31 state words + word 31 = f | b<<8  (glibc TYPE_3)
__device__ inline uint32_t rand_next(uint32_t* r) {
    int f = r[31] & 0xFF, b = (r[31] >> 8) & 0xFF;
    uint32_t val = r[f] + r[b];
    r[f] = val;
    f += 1;
    if (f >= 31) { f = 0; b += 1; }
    else { b += 1; if (b >= 31) b = 0; }
    r[31] = (uint32_t)(f | (b << 8));
    return val >> 1;
}

__device__ inline int unique_child_obs(int index, const int32_t* child, const float* score, const int32_t* n_to_o,
                                       int32_t* cn, int32_t* co) {
    int n = 0;
    for (int i = 0; i < 7; ++i) {
        int c = child[(size_t)index * 7 + i];
        if (c == 0) continue;
        int o = n_to_o[c];
        int found = -1;
        for (int j = 0; j < n; ++j)
            if (co[j] == o) { found = j; break; }
        if (found < 0) { cn[n] = c; co[n] = o; n += 1; }
        else if (score[c] > score[cn[found]]) cn[found] = c;
    }
    return n;
}

__global__ void k_select(int B, int N, const int32_t* roots, const int32_t* child, const int32_t* visit,
                         const float* value, const float* variance, const float* score, const int32_t* n_to_o, int low,
                         uint32_t* rng, const float* nq_table, int nq_size, int32_t* trace, int32_t* trace_len,
                         int max_trace) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    size_t off = (size_t)b * N;
    child += off * 7; visit += off; value += off; variance += off; score += off; n_to_o += off;
    uint32_t* r = rng + (size_t)b * 32;
    int32_t* tr = trace + (size_t)b * max_trace;
    int index = roots[b], len = 0;
    int32_t cn[7], co[7];
    for (;;) {
        if (len >= max_trace) { len = -1; break; }
        tr[len++] = index;
        int n = unique_child_obs(index, child, score, n_to_o, cn, co);
        if (n == 0) break;
        int32_t lows[7];
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (visit[co[i]] < low) lows[m++] = co[i];
        if (m > 0) {
            int o = lows[rand_next(r) % (uint32_t)m];
            for (int i = 0; i < n; ++i)
                if (co[i] == o) { index = cn[i]; break; }
        } else {
            int total = 0;
            for (int i = 0; i < n; ++i) total += visit[co[i]];
            float coeff = nq(nq_table, nq_size, total);
            int best = 0;
            float max_q = 0;
            for (int i = 0; i < n; ++i) {
                float t1 = value[co[i]] + score[cn[i]];
                float val = t1 - score[index];
                float ratio = variance[co[i]] / (float)visit[co[i]];
                float root = sqrtf(ratio);
                float prod = coeff * root;
                float q = val + prod;
                if (i == 0) max_q = q;
                else if (q > max_q) { max_q = q; best = i; }
            }
            index = cn[best];
        }
    }
    trace_len[b] = len;
}

__device__ inline void backup_one(const int32_t* tr, int len, int32_t* visit, float* value, float* variance,
                                  const int32_t* n_to_o, const float* score, double _value, double _variance,
                                  double gamma) {
    for (int i = len - 1; i >= 0; --i) {
        int idx = tr[i];
        _value = _value - (double)score[idx];
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

// core.h:262-301
__device__ inline void backup_one_mixture(const int32_t* tr, int len, int32_t* visit, float* value, float* variance,
                                          const int32_t* n_to_o, const float* score, double _value, double _variance,
                                          double gamma) {
    for (int i = len - 1; i >= 0; --i) {
        int idx = tr[i];
        _value = _value - (double)score[idx];
        int o = n_to_o[idx];
        visit[o] += 1;
        double a = _value * _value;
        double b = (double)(value[o] * value[o]);   // float*float, rounded to float first
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
        _variance = _variance * (gamma * gamma);
    }
}

__global__ void k_backup(int B, int N, const int32_t* trace, const int32_t* trace_len, int max_trace, int32_t* visit,
                         float* value, float* variance, const int32_t* n_to_o, const float* score, const double* _value,
                         const double* _variance, double gamma) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    size_t off = (size_t)b * N;
    backup_one(trace + (size_t)b * max_trace, trace_len[b], visit + off, value + off, variance + off, n_to_o + off,
               score + off, _value[b], _variance[b], gamma);
}

__global__ void k_backup_lp(int B, int N, const int32_t* trace, const int32_t* trace_len, int max_trace, int32_t* visit,
                            float* value, float* variance, const int32_t* n_to_o, const float* score, const uint8_t* end,
                            const int32_t* _child, const int32_t* _obs, const int32_t* kk, const float* _value,
                            const float* _variance, double gamma, int mixture, int averaged) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    size_t off = (size_t)b * N;
    visit += off; value += off; variance += off; n_to_o += off; score += off; end += off;
    const int32_t* tr = trace + (size_t)b * max_trace;
    int len = trace_len[b], k = kk[b];
    auto backup = [&](double v0, double var0) {
        if (mixture) backup_one_mixture(tr, len, visit, value, variance, n_to_o, score, v0, var0, gamma);
        else backup_one(tr, len, visit, value, variance, n_to_o, score, v0, var0, gamma);
    };
    if (k > 0) {
        double v_tmp = 0, var_tmp = 0;
        for (int i = 0; i < k; ++i) {
            int c = _child[b * 7 + i], o = _obs[b * 7 + i];
            if (visit[o] == 0) {
                visit[o] += 1;
                if (end[c]) { value[o] = 0; variance[o] = 0; }
                else { value[o] = _value[b * 7 + i]; variance[o] = _variance[b * 7 + i]; }
            }
            if (averaged) {
                double gv = gamma * (double)value[o];
                v_tmp = v_tmp + ((double)score[c] + gv);
                var_tmp = var_tmp + (double)variance[o];
            } else {
                double gs = gamma * (double)score[c];
                double gg = gamma * gamma;
                backup((double)value[o] + gs, gg * (double)variance[o]);
            }
        }
        if (averaged) {
            v_tmp = v_tmp / (double)k;
            var_tmp = var_tmp * (gamma * gamma / (double)k);
            backup(v_tmp, var_tmp);
        }
    } else {
        backup((double)score[tr[len - 1]], 0);
    }
}

__global__ void k_unique(int B, int N, const int32_t* index, const int32_t* child, const float* score,
                         const int32_t* n_to_o, int32_t* c_nodes, int32_t* c_obs, int32_t* count) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    size_t off = (size_t)b * N;
    int32_t cn[7] = {0, 0, 0, 0, 0, 0, 0}, co[7] = {0, 0, 0, 0, 0, 0, 0};
    int n = unique_child_obs(index[b], child + off * 7, score + off, n_to_o + off, cn, co);
    for (int i = 0; i < 7; ++i) { c_nodes[b * 7 + i] = cn[i]; c_obs[b * 7 + i] = co[i]; }
    count[b] = n;
}

__global__ void k_all_childs(int B, int N, const int32_t* roots, const int32_t* child, uint8_t* mark, int32_t* queue) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    size_t off = (size_t)b * N;
    child += off * 7; mark += off; queue += off;
    for (int i = 0; i < N; ++i) mark[i] = 0;
    int head = 0, tail = 0;
    queue[tail++] = roots[b];
    mark[roots[b]] = 1;
    while (head < tail) {
        int n = queue[head++];
        for (int a = 0; a < 7; ++a) {
            int c = child[(size_t)n * 7 + a];
            if (!mark[c]) { mark[c] = 1; queue[tail++] = c; }
        }
    }
}

}  // namespace tmcts_core

using namespace tmcts_core;
#define GRID(B) dim3(((B) + 63) / 64), dim3(64), 0, (hipStream_t)stream

// ---------------------------------------------------------------------------------------------------
// The reference's distribution helpers (agents/cppmodule/core.h:387-449: transform_distribution, mean_variance_dist /
// mean_dist; BASELINE configs[4]), one lane per distribution, bins in the reference's order so float / double roundings
// are the reference's.  Mass that falls beyond the last bin is dropped (the reference writes result[bins] there, one
// float past its vector).
// ---------------------------------------------------------------------------------------------------
__global__ void k_dist_transform(int n, int bins, const float* __restrict__ dist, double vmin, double vmax,
                                 const double* __restrict__ shift, double scale, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* d = dist + (size_t)i * bins;
    float* r = out + (size_t)i * bins;
    for (int b = 0; b < bins; ++b) r[b] = 0.0f;
    const double delta = (vmax - vmin) / bins;
    const double bin_shift = shift[i] / delta;
    for (int b = 0; b < bins; ++b) {
        const double lb = fmax(b * scale + bin_shift, 0.);
        const int b_lb = (int)floor(lb);
        const double ub = fmin(lb + scale, (double)bins);
        const int b_ub = (int)floor(ub);
        const double frac = b_ub - lb;
        if (b_lb < bins) r[b_lb] = (float)((double)r[b_lb] + (double)d[b] * frac);
        if (b_ub < bins) r[b_ub] = (float)((double)r[b_ub] + (double)d[b] * (1 - frac));
    }
}
__global__ void k_dist_mean_variance(int n, int bins, const float* __restrict__ dist, double vmin, double vmax,
                                     double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* d = dist + (size_t)i * bins;
    const double delta = (vmax - vmin) / bins;
    double mean = 0, m2 = 0, center = vmin + 0.5 * delta;
    for (int b = 0; b < bins; ++b) {
        const double tmp = center * d[b];
        mean += tmp;
        m2 += center * tmp;
        center += delta;
    }
    out[2 * i] = mean;
    out[2 * i + 1] = m2 - mean * mean;
}

// ---------------------------------------------------------------------------------------------------
// agents/core_distributional.py (the numba kernels of the reference's unfinished distributional agent): shift_distribution
// :12-37, policy_dist :66-79, backup_trace_distributional :108-124, batched over trees, one lane per tree.  The reference
// code is `fastmath` numba: no bit pattern is defined, the arithmetic here follows numba's typing (float32 arrays,
// float64 scalars; see oracle/dist_oracle.c) and is held to a float tolerance against a pure-Python run of the reference.
// ---------------------------------------------------------------------------------------------------
__device__ inline int py_index(int i, int n) { return i < 0 ? i + n : i; }   // Python / numba wraparound indexing
__device__ inline void distpy_shift(const float* d, int bins, double x, double vmin, double vmax, float* r) {
    const double delta = (vmax - vmin) / bins;
    for (int b = 0; b < bins; ++b) r[b] = 0.0f;
    const double bin_shift = x / delta;
    const double fraction = bin_shift - floor(bin_shift);
    for (int b = 0; b < bins; ++b) {
        int b_lb = (int)(b + bin_shift);
        if (b_lb >= bins) b_lb = bins - 1;
        const int b_ub = (b_lb + 1 >= bins) ? bins - 1 : b_lb + 1;
        const int lo = py_index(b_lb, bins), hi = py_index(b_ub, bins);
        if (lo >= 0) r[lo] = (float)((double)r[lo] + (double)d[b] * (1 - fraction));
        if (hi >= 0) r[hi] = (float)((double)r[hi] + (double)d[b] * fraction);
    }
}
__global__ void k_distpy_shift(int n, int bins, const float* __restrict__ dist, const double* __restrict__ x, double vmin,
                               double vmax, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    distpy_shift(dist + (size_t)i * bins, bins, x[i], vmin, vmax, out + (size_t)i * bins);
}
__global__ void k_distpy_policy(int B, int N, const int32_t* __restrict__ child_nodes, const int32_t* __restrict__ n_child,
                                const float* __restrict__ node_stats, const double* __restrict__ curr_reward,
                                int32_t* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    const int32_t* cn = child_nodes + (size_t)t * 7;
    const float* st = node_stats + (size_t)t * N * 5;
    const int nc = min(max(n_child[t], 0), 7);
    if (nc == 0) { out[t] = 0; return; }
    const double eps = 1e-3;
    double n = 0;
    float s0[7], s1[7];
    for (int i = 0; i < nc; ++i) {
        const float* ns = st + (size_t)cn[i] * 5;
        n += ns[0];
        const float a = ns[1] + ns[2];
        s0[i] = (float)((double)a - curr_reward[t]);
        s1[i] = (float)((double)ns[3] / ((double)ns[0] + eps));
    }
    const double alpha = 1 - 1 / n;
    const double coeff = 10 * log(1 - log(-log(alpha) / log(2.0)) / log(22.0)) / log(41.0);
    int best = 0;
    double best_q = 0;
    for (int i = 0; i < nc; ++i) {
        const double q = (double)s0[i] + coeff * (double)sqrtf(s1[i]);
        if (i == 0) { best_q = q; continue; }
        if (best_q != best_q) break;                       // np.argmax: the first NaN wins
        if (q != q || q > best_q) { best_q = q; best = i; }
    }
    out[t] = cn[best];
}
__global__ void k_distpy_backup(int B, int N, int bins, const int32_t* __restrict__ trace, const int32_t* __restrict__ trace_len,
                                int max_trace, float* __restrict__ node_stats, float* __restrict__ node_dist,
                                const double* __restrict__ r, const float* __restrict__ leaf_dist, double vmin, double vmax,
                                float* __restrict__ scratch) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    const float* dist = leaf_dist + (size_t)t * bins;
    float* sc = scratch + (size_t)t * bins;
    const double delta = (vmax - vmin) / bins;
    double mean = 0;
    for (int b = 0; b < bins; ++b) mean += (double)dist[b] * ((b + 0.5) * delta);
    const int len = trace_len[t];
    for (int k = 0; k < len; ++k) {
        const int idx = trace[(size_t)t * max_trace + k];
        float* ns = node_stats + ((size_t)t * N + idx) * 5;
        float* nd = node_dist + ((size_t)t * N + idx) * bins;
        const double _r = r[t] - (double)ns[2];
        distpy_shift(dist, bins, _r, vmin, vmax, sc);
        const float n0 = ns[0];
        for (int b = 0; b < bins; ++b) {
            const float m = nd[b] * n0;
            const float u = m + sc[b];
            nd[b] = (float)((double)u / ((double)n0 + 1.0));
        }
        const double x = mean + _r;
        const float n1 = n0 + 1.0f;
        ns[0] = n1;
        const double d1 = x - (double)ns[1];
        const float v = (float)((double)ns[1] + d1 / (double)n1);
        ns[1] = v;
        const double d2 = x - (double)v;
        const float m2 = (float)((double)ns[4] + d1 * d2);
        ns[4] = m2;
        if (n1 > 1.0f) ns[3] = (float)((double)m2 / ((double)n1 - 1.0));
    }
}

extern "C" {
int tm_core_select_trace_obs(int B, int N, const int32_t* roots, const int32_t* child, const int32_t* visit,
                             const float* value, const float* variance, const float* score, const int32_t* n_to_o,
                             int low, uint32_t* rng, const float* nq_table, int nq_size, int32_t* trace,
                             int32_t* trace_len, int max_trace, void* stream) {
    hipLaunchKernelGGL(k_select, GRID(B), B, N, roots, child, visit, value, variance, score, n_to_o, low, rng, nq_table,
                       nq_size, trace, trace_len, max_trace);
    return (int)hipGetLastError();
}
int tm_core_backup_trace_obs(int B, int N, const int32_t* trace, const int32_t* trace_len, int max_trace, int32_t* visit,
                             float* value, float* variance, const int32_t* n_to_o, const float* score,
                             const double* _value, const double* _variance, double gamma, void* stream) {
    hipLaunchKernelGGL(k_backup, GRID(B), B, N, trace, trace_len, max_trace, visit, value, variance, n_to_o, score,
                       _value, _variance, gamma);
    return (int)hipGetLastError();
}
int tm_core_backup_trace_obs_lp(int B, int N, const int32_t* trace, const int32_t* trace_len, int max_trace,
                                int32_t* visit, float* value, float* variance, const int32_t* n_to_o, const float* score,
                                const uint8_t* end, const int32_t* _child, const int32_t* _obs, const int32_t* k,
                                const float* _value, const float* _variance, double gamma, int mixture, int averaged,
                                void* stream) {
    hipLaunchKernelGGL(k_backup_lp, GRID(B), B, N, trace, trace_len, max_trace, visit, value, variance, n_to_o, score,
                       end, _child, _obs, k, _value, _variance, gamma, mixture, averaged);
    return (int)hipGetLastError();
}
int tm_core_get_unique_child_obs(int B, int N, const int32_t* index, const int32_t* child, const float* score,
                                 const int32_t* n_to_o, int32_t* c_nodes, int32_t* c_obs, int32_t* count, void* stream) {
    hipLaunchKernelGGL(k_unique, GRID(B), B, N, index, child, score, n_to_o, c_nodes, c_obs, count);
    return (int)hipGetLastError();
}
int tm_core_get_all_childs(int B, int N, const int32_t* roots, const int32_t* child, uint8_t* mark, int32_t* queue,
                           void* stream) {
    hipLaunchKernelGGL(k_all_childs, GRID(B), B, N, roots, child, mark, queue);
    return (int)hipGetLastError();
}
int tm_dist_transform(int n, int bins, const float* dist, double vmin, double vmax, const double* shift, double scale,
                      float* out, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_dist_transform, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, bins, dist, vmin, vmax,
                       shift, scale, out);
    return (int)hipGetLastError();
}
int tm_distpy_shift(int n, int bins, const float* dist, const double* x, double vmin, double vmax, float* out, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_distpy_shift, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, bins, dist, x, vmin, vmax, out);
    return (int)hipGetLastError();
}
int tm_distpy_policy(int B, int n_nodes, const int32_t* child_nodes, const int32_t* n_child, const float* node_stats,
                     const double* curr_reward, int32_t* out, void* stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_distpy_policy, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, n_nodes, child_nodes, n_child,
                       node_stats, curr_reward, out);
    return (int)hipGetLastError();
}
int tm_distpy_backup(int B, int n_nodes, int bins, const int32_t* trace, const int32_t* trace_len, int max_trace,
                     float* node_stats, float* node_dist, const double* r, const float* leaf_dist, double vmin, double vmax,
                     float* scratch, void* stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_distpy_backup, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, n_nodes, bins, trace,
                       trace_len, max_trace, node_stats, node_dist, r, leaf_dist, vmin, vmax, scratch);
    return (int)hipGetLastError();
}
int tm_dist_mean_variance(int n, int bins, const float* dist, double vmin, double vmax, double* out, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_dist_mean_variance, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, bins, dist, vmin,
                       vmax, out);
    return (int)hipGetLastError();
}
}
