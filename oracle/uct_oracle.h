/*
 * ORACLE — test infrastructure only (see oracle/tetris_engine.h).
 * C restatement of the reference's native tree kernels, agents/cppmodule/core.h and special.h.
 * PINNED: tests/test_oracle_uct.py checks every function here against the reference's own
 * core.cpp compiled in place (oracle/_ref/core*.so) on seeded random DAGs, bit for bit,
 * including the libc rand() stream.
 */
#ifndef UCT_ORACLE_H
#define UCT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NACT 7

/* glibc rand()/srand() (TYPE_3 additive feedback generator, r[i] = r[i-3] + r[i-31]). */
typedef struct {
    int32_t r[34];
    int32_t f, b; /* front / rear index */
} orc_rand_t;
void orc_srand(orc_rand_t *s, uint32_t seed);
int32_t orc_rand(orc_rand_t *s);

/* special.h:26-33 */
double orc_norm_quantile(double t);

/* core.h:111-144 ; returns number of unique children (<=7) */
int orc_get_unique_child_obs(int index, const int32_t *child, const float *score, const int32_t *n_to_o,
                             int32_t *c_nodes, int32_t *c_obs);
/* core.h:65-77 */
int orc_check_low(const int32_t *c_obs, int n, const int32_t *visit, int low, orc_rand_t *rng);
/* core.h:83-105 */
int orc_policy_clt(const int32_t *nodes, const int32_t *visit, const float *value, const float *variance, int n);
/* core.h:167-224 ; returns trace length, or -1 when it exceeds cap */
int orc_select_trace_obs(int index, const int32_t *child, const int32_t *visit, const float *value,
                         const float *variance, const float *score, const int32_t *n_to_o, int low,
                         orc_rand_t *rng, int32_t *trace, int cap);
/* core.h:226-260 */
void orc_backup_trace_obs(const int32_t *trace, int len, int32_t *visit, float *value, float *variance,
                          const int32_t *n_to_o, const float *score, double _value, double _variance, double gamma);
/* core.h:262-301 */
void orc_backup_trace_mixture_obs(const int32_t *trace, int len, int32_t *visit, float *value, float *variance,
                                  const int32_t *n_to_o, const float *score, double _value, double _variance,
                                  double gamma);
/* core.h:303-381 (the function the Python agents call) */
void orc_backup_trace_obs_LP(const int32_t *trace, int len, int32_t *visit, float *value, float *variance,
                             const int32_t *n_to_o, const float *score, const uint8_t *end, const int32_t *_child,
                             const int32_t *_obs, int k, const float *_value, const float *_variance, double gamma,
                             int mixture, int averaged);
/* agent.cpp:496-566 (the all-C++ agent's twin: float carry, end_obs[o], no gamma^2) */
void orc_backup_obs_single_cppagent(const int32_t *trace, int len, int32_t *visit, float *value, float *variance,
                                    const int32_t *n_to_o, const float *score, float _val, float _var, double gamma);
void orc_backup_obs_cppagent(const int32_t *trace, int len, int32_t *visit, float *value, float *variance,
                             const int32_t *n_to_o, const float *score, const uint8_t *end_obs,
                             const int32_t *_child, const int32_t *_obs, int k, const float *_value,
                             const float *_variance, double gamma, float leaf_score);
/* core.h:32-50 ; mark[i]=1 for every reachable node (0 is always followed); returns count */
int orc_get_all_childs(int index, const int32_t *child, int n_nodes, uint8_t *mark);
/* the same set by the device collector's schedule (descending sweeps over a pending bitmap, tree.hip gc_sweep_mark) */
int orc_sweep_marks(int index, const int32_t *child, int n_nodes, uint8_t *mark, int seg_nodes, long *stats);

#ifdef __cplusplus
}
#endif
#endif
