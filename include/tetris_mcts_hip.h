/*
 * tetris_mcts_hip.h — C ABI of libtetris_mcts_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the reference's hot path:
 *   - environment step            : pyTetris `Tetris.play/copy_from/getState/reset`
 *                                   (call sites play.py:150,165; agents/agent.py:101-129,140-145)
 *   - tree agent (select/expand/backup, transposition + observation tables)
 *                                 : agents/cppmodule/agent.cpp TreeAgent/MCTSAgent (agent.cpp:84-568) and the
 *                                   Python twins agents/agent.py:90-193, ValueSim.py:52-99, ValueSimLP.py:13-70
 *   - the five `agents.cppmodule.core` functions (core.cpp:20-26) in batched form over device arrays
 *   - leaf evaluator              : model/model_vv.py:13-52,210-217
 * All pointers are DEVICE pointers unless named host_*.  Every function enqueues work on `stream`
 * (a hipStream_t passed as void*) and returns a hipError_t value (0 = success); nothing synchronises.
 * No torch types cross this boundary.  One process drives one GPU; games shard across processes.
 */
#ifndef TETRIS_MCTS_HIP_H
#define TETRIS_MCTS_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TM_NACT 7
#define TM_GAME_DW 16      /* packed game, ENGINE_SPEC.md section 2 (64 bytes) */
#define TM_OBS_DW 12       /* packed observation, ENGINE_SPEC.md section 7 (48 bytes) */
#define TM_REC_DW 32       /* node record (128 bytes = one cache line): eight 16-byte pieces.  Pieces 0..6 = the node's unique
                              children in selection order: (child, obs, child score bits, own score bits); piece 7 =
                              (TM_REC_PCHILD, TM_REC_OBS, TM_REC_SCORE, TM_REC_HDR); see DESIGN.md "node store" */
#define TM_REC_PCHILD 28   /* the child the last walk through this node descended into (0: none yet): the walk prefetches it */
#define TM_REC_OBS 29      /* node_to_obs */
#define TM_REC_SCORE 30    /* float32 score of the node's game */
#define TM_REC_HDR 31      /* bits 0-2 number of unique children, bit 24 game ended, bit 25 expanded */
#define TM_KIDS_DW 8       /* raw children row: child[7] in action order + the node's observation */
#define TM_GS_DW 64        /* per-game control block */
#define TM_LEAF_DW 32      /* per-game leaf hand-off between the front and back halves of a simulation */
#define TM_DIST_ROW 64       /* floats per row of node_dist / eval_dist (dist_bins <= 64 atoms, zero padded) */
#define TM_GC_PART_DW 192   /* per-game scratch of a collection: (free nodes, free observations, harvested tuples) per collector workgroup */
#define TM_VALUENET_PARAMS 478342
#define TM_VALUENET_SCRATCH 9728       /* floats of scratch per state, tm_valuenet_forward_plain */
#define TM_VALUENET_SCRATCH_MFMA 2064  /* floats of scratch per state, tm_valuenet_forward / _requests; no initial contents required (the
                                          kernels keep a counter per tile of 32 or 64 states in the rows' padding: every evaluation
                                          clears them first) */
#define TM_VALUENET_PREPARED 477184    /* floats: conv2 + conv3 + fc1 operand streams */
#define TM_DISTNET_PARAMS(atoms) (279232 + 129 * (atoms))  /* floats: conv1.w[32][1][4][4] conv1.b[32] conv2.w[32][32][4][4] conv2.b[32]
                                          fc1.w[128][2048] fc1.b[128] fc_v.w[atoms][128] fc_v.b[atoms] (model/model_distributional.py:33-42) */
#define TM_DISTNET_PARAMS_50 285682    /* TM_DISTNET_PARAMS(50) */
#define TM_DISTNET_PREPARED 278528     /* floats: conv2 + fc1 operand streams */
#define TM_DISTNET_SCRATCH 2048        /* floats of scratch per state (conv2's output), tm_distnet_forward / _requests */

/* per-game control block (int32 words) */
enum {
    TM_GS_ROOT = 0, TM_GS_EPISODE, TM_GS_NFREE_NODE, TM_GS_NFREE_OBS, TM_GS_TRACE_LEN, TM_GS_PENDING,
    TM_GS_ERR, TM_GS_N_EXPAND, TM_GS_N_SIMS, TM_GS_N_GC, TM_GS_RNG_POS, TM_GS_N_NQ_FALLBACK,
    TM_GS_LEAF, TM_GS_LEAF_END, TM_GS_K_EVAL, TM_GS_LEAF_SCORE,
    TM_GS_TRACE_SUM, /* sum of trace lengths over all simulations (for bytes/simulation accounting) */
    TM_GS_N_EVAL,    /* leaf states handed to the evaluator (requests posted) */
    TM_GS_N_POOL_RESET, /* tm_pool_reset calls that hit this game */
    TM_GS_MAX_TRACE,    /* longest trace of any simulation so far */
    TM_GS_CYC_BACK = 20, TM_GS_CYC_SELECT, TM_GS_CYC_EXPAND, /* shader cycles of the last simulation's phases */
    TM_GS_CYC_TAIL,      /* duration of the last GC in units of 16 cycles; +1: nodes reachable at that GC */
    TM_GS_N_DROPPED = 25, /* replay tuples a GC could not store because the harvest buffer was full (drain it more often) */
    TM_GS_LOW_NODE = 26, TM_GS_LOW_OBS,  /* lowest node / observation index ever allocated (GC skips untouched entries) */
    TM_GS_FIRST_MISS = 28, /* levels of the last walk that were taken over from the walk before it (verified in parallel, tree.hip) */
    TM_GS_PREFIX_SUM,      /* sum of TM_GS_FIRST_MISS over all simulations */
    /* garbage collection by the collector workgroups of tm_sim_step (a game that collects does not simulate; tree.hip) */
    TM_GS_GC_PHASE = 32, /* 0 none; (launch << 4) | 1 requested; 2 marking, 3 counting + re-inserting the kept nodes, 4 writing the free
                            lists + re-inserting the kept observations; (launch << 4) | 7 complete (the game resumes in a later
                            launch); speculative marking while the game simulates: (launch << 4) | 8 requested, 9 under way,
                            (launch << 4) | 10 the pool ran dry meanwhile */
    TM_GS_GC_ARRIVE,     /* collector workgroups that have done their share of the current step (bits 8..: some left work) */
    TM_GS_GC_ACTIVE4,    /* in the control block of every FOURTH game (game 4k of the store or slice - slices begin at multiples of four
                            games): bit j = game 4k + j's phase word may be non-zero.  The collector workgroups of a launch look at
                            these words and read the phase words they point at, instead of every game's (set with the request by the
                            game's wave, cleared by it when it resumes; a bit without a phase is harmless, a phase without its bit
                            would never be served) */
    TM_GS_GC_RSV1,       /* (unused) */
    TM_GS_GC_WORK,       /* marking: chunks were flagged when the last launch that looked ended (the next one marks) */
    TM_GS_GC_MARK_LAUNCHES, /* launches in which a collector workgroup marked for this game (all collections) */
    TM_GS_GC_SLICES,     /* launches in which collector workgroups worked on a collection of this game (all collections) */
    TM_GS_GC_RETRY,      /* the suspended expansion has already been through a collection */
    /* per-move simulation quota: tm_move_begin adds `sims` to the target; a launch starts a simulation for a game only
       while started < target, so games that lost launches to a collection catch up in extra launches (tm_sims_remaining) */
    TM_GS_SIM_TARGET = 40, TM_GS_SIM_STARTED,
    TM_GS_CYC_VERIFY,    /* shader cycles of the last walk's parallel prefix verification (part of TM_GS_CYC_SELECT) */
    TM_GS_N_WALK_MISS,   /* tree levels at which the walk descended into another child than the predicted one (all simulations) */
    TM_GS_POOL_FULL,     /* the reachable tree fills the pool (TM_ERR_POOL): no collection is attempted until the root moves */
    TM_GS_GC_IN_MOVE,    /* a collection has completed since the root last moved: until it moves again nothing becomes unreachable
                            (the search only adds links), so the next exhaustion is counted as the reference's collection that
                            frees nothing (agents/agent.py:96-97) and goes straight to TM_GS_POOL_FULL */
    TM_GS_GC_REQ_AT,     /* the launch in which the game asked for the collection it is waiting for (the waiting games' steps that
                            do not all fit a launch are served oldest request first) */
    TM_GS_LEAF_OBS,      /* the pending leaf's observation (TM_SIM_EVAL_NEEDED: where the backup files the evaluator's output) */
    TM_GS_N_EVAL_SKIP,   /* leaf-parallel kinds: unique children of expanded leaves whose evaluation the backup would discard
                            (observation already visited / finished) - not posted under TM_SIM_EVAL_NEEDED, counted always */
    TM_GS_N_EVAL_CACHED, /* TM_KIND_VALUESIM / TM_KIND_CPPAGENT under TM_SIM_EVAL_NEEDED: leaves answered from obs_eval */
    TM_GS_GC_NGC,        /* collector workgroups of the launch that began the collection under way (its shares are cut for that many) */
    TM_GS_GC_BLOCKS,     /* the marker's work, all collections: blocks of child rows loaded (tree.hip GC_BLOCK_NODES nodes each), */
    TM_GS_GC_ITERS,      /* ... rounds of its block-local marking (one workgroup barrier each), */
    TM_GS_GC_MARK_CYC,   /* ... shader cycles / 64 it was at it (summed over its workgroups) */
    TM_GS_GC_FLAGS       /* 4 words = 128 flags, the marking's work list: chunks of the index range (tree.hip gc_chunk_log2) that hold
                            pending nodes - marked, children not looked at yet (device-scope atomics: a marking workgroup takes the
                            flags of its share of the range and returns what it leaves, the other marking workgroups of the game and
                            the write barrier of a speculative marking raise them) */
    , TM_GS_GC_MARK_PARTS = 58, /* sum over the marking launches of the marking workgroups the game had (each owns a share of its range) */
    TM_GS_GC_MARK_SHARED,      /* sum over the marking launches of the games its marking workgroup had in that launch (its time is shared) */
    TM_GS_GC_CYC_LOAD,         /* the marker's waves, shader cycles / 64: waiting for a block's rows, */
    TM_GS_GC_CYC_ROUNDS,       /* ... marking inside a block, */
    TM_GS_GC_CYC_WAVES,        /* ... in their loop all in all (the rest: looking for work, waiting for it), */
    TM_GS_GC_IDLE_TURNS        /* turns of that loop that found nothing flagged */
};
#define TM_GC_FLAG_WORDS 4
/* error bits in TM_GS_ERR */
#define TM_ERR_POOL 1      /* node pool exhausted even after reclaiming unreachable nodes */
#define TM_ERR_TRACE 2     /* trace longer than max_trace */
#define TM_ERR_TABLE 4     /* transposition table full */
#define TM_ERR_EVAL_LIST 8 /* the dense request list overflowed: the caller did not flip eval_parity between two TM_SIM_FRONT launches */
#define TM_ERR_GC_GRID 16  /* a collection under way was met by a launch with another number of collector workgroups than the launch
                              that began it (launches over another range of games: a sub-batch, the full store): the shares and the
                              arrival count would not add up, so the collection is not touched and the game is flagged */

/* agent numerics (which reference twin is reproduced bit for bit) */
#define TM_KIND_VALUESIM 0     /* agents/ValueSim.py:76-94      : evaluate the leaf, fp64 carry             */
#define TM_KIND_VALUESIM_LP 1  /* agents/ValueSimLP.py:44-70    : evaluate the leaf's unique children, core.h:303-381 */
#define TM_KIND_CPPAGENT_LP 2  /* agent.cpp:420-436,517-566     : float carry, end_obs[o], no gamma^2       */
#define TM_KIND_CPPAGENT 3     /* agent.cpp:437-446,496-513     : float carry, single leaf                 */
#define TM_KIND_VANILLA 4      /* agents/Vanilla.py:42-64       : random rollout to the end of the game (CPython MT19937 randint), variance 1e3 */
#define TM_KIND_VANILLA_C 5    /* agents/VanillaC.py:5-14 via agent.cpp:447-455 (evaluator type 1): rollout with randint(0, 7),
                                  variance 1e5, float carry, the C++ agent's root statistics */
#define TM_KIND_DIST 6         /* DistValueSim: the reference's unfinished distributional agent (agents/DistValueSimOnline.py, does
                                  not import) rebuilt from its working parts - agents/core_distributional.py:12-124 - on a tree without
                                  the observation projection: obs_stat[node] = (visit, mean, variance, M2) as floats, node_dist[node] =
                                  dist_bins atoms over [dist_vmin, dist_vmax); the evaluator fills eval_dist; low = 5 */

typedef struct tm_store {
    /* sizes */
    int32_t n_games;      /* games owned by this process (one wavefront each) */
    int32_t max_nodes;    /* node / observation pool size per game (agents/agent.py:36, ValueSim.py:16) */
    int32_t table_cap;    /* transposition table slots per game, power of two >= 2*max_nodes */
    int32_t max_trace;    /* trace capacity per game */
    int32_t eval_slots;   /* leaf-evaluation slots per game: 1 (ValueSim) or 7 (leaf-parallel) */
    int32_t nq_size;      /* entries in nq_table */
    /* environment (pyTetris ctor arguments, play.py:75) */
    int32_t app, scoring, randomizer;
    /* search */
    int32_t low;          /* check_low threshold (core.h:65-77): 1 for ValueSim/LP, 5 for Vanilla */
    int32_t kind;         /* TM_KIND_* */
    int32_t min_visits_to_store; /* ValueSim.py:14 / ValueSimLP.py:11 */
    int32_t online;       /* harvest replay tuples on GC (ValueSim.py:109-115) */
    int32_t replay_cap;   /* capacity (tuples) of the replay buffer */
    int32_t gc_slice_cycles; /* shader cycles the collector workgroups of a tm_sim_step launch may spend marking (0: no limit) */
    double gamma;
    /* node store, per game contiguous */
    uint32_t *node_rec;   /* [G][N][32] 7 x (child, obs, score, own score) unique children in selection order, then (0, self_obs, self_score, hdr) */
    uint32_t *node_game;  /* [G][N][16] packed game */
    uint32_t *obs_stat;   /* [G][N][4]  visit (bits 0-30) | end << 31, value(f32), variance(f32), sqrtf(variance / (float)visit)
                             (the exploration term of policy_clt, core.h:98, evaluated when the statistics change) */
    uint32_t *obs_key;    /* [G][N][12] packed observation */
    uint64_t *node_tab;   /* [G][cap]   (hash>>32)<<32 | node index ; index 0 = empty */
    uint64_t *obs_tab;    /* [G][cap] */
    int32_t *free_node;   /* [G][N] free-index stacks (pop from the end, agents/agent.py:99) */
    int32_t *free_obs;    /* [G][N] */
    int32_t *gs;          /* [G][TM_GS_DW] control blocks */
    uint32_t *rng;        /* [G][32] glibc rand() state (31 words) per game (core.h:62,76) */
    uint32_t *env_game;   /* [G][16] the real games */
    int32_t *env_line_stats; /* [G][4] */
    uint32_t *trace;      /* [G][max_trace][4] the last walk: per level (node, its observation, its score bits, its record's header) */
    int32_t *leaf;        /* [G][32] unique children of the leaf: node[7], obs[7], score bits[7], end[7] */
    int32_t *eval_obs;    /* [G*eval_slots] observation index inside the game's pool, 0 = unused slot */
    float *eval_v;        /* [G*eval_slots] evaluator outputs */
    float *eval_var;
    const float *nq_table;/* [nq_size] (float)norm_quantile(n), special.h:26-33, built on the host with libm */
    uint8_t *gc_mark;     /* [G][2][N/8 rounded to 16B] scratch bitmaps for GC: reachable nodes, observations of reachable nodes */
    int32_t *gc_queue;    /* [G][N] queue scratch of the one-group collection (update_root's exhausting pop, the single calls) */
    /* replay tuples harvested at GC (ValueSim.py:122-159): per game ring, gathered by the host side */
    uint32_t *replay_obs; /* [G][replay_cap][12] */
    float *replay_stat;   /* [G][replay_cap][4] value, variance, visit, 0 */
    int32_t *replay_count;/* [G] */
    uint32_t *mt_state;   /* [G][625] CPython random state per game (624 words + index), TM_KIND_VANILLA / TM_KIND_VANILLA_C rollouts (Vanilla.py:4,52) */
    uint32_t *node_child; /* [G][N][8] raw child[7] per action (agents/agent.py:61) + the node's own observation (node_to_obs once more:
                             the collectors' marker reads children and observation in one 32-byte row) */
    int32_t *gc_part;     /* [G][TM_GC_PART_DW] per-collector counts of a collection in progress */
    /* TM_KIND_DIST only */
    float *node_dist;     /* [G][N][TM_DIST_ROW] the nodes' value distributions (agents/core_distributional.py node_dist) */
    float *eval_dist;     /* [G][TM_DIST_ROW] the evaluator's distribution for the game's pending leaf (model_distributional.Net) */
    const double *nq_table_d; /* [nq_size] norm_quantile(n) in double (policy_dist multiplies in double), host libm */
    double dist_vmin, dist_vmax;
    int32_t dist_bins;
    int32_t gc_spec_nodes; /* a game with fewer free nodes than this has its tree marked speculatively while it goes on simulating
                              (tree.hip GC_SPEC_*); 0: never */
    /* dense evaluation requests: beside the per-game slots (eval_obs) every tm_sim_step launch appends the requests it posts to
       a list the built-in evaluators draw dense work from (conv waves, 32-row FC tiles).  TM_EVAL_SEGS(n_games) segments (game
       g appends to segment g % segs: one device atomic per game and launch, spread over the segments' counters); entry d of
       segment s lives at eval_list[(s + segs * (d / eval_slots)) * eval_slots + d % eval_slots].  Two sets of counters: a launch
       with TM_SIM_FRONT appends under eval_parity and clears the other set, so the CALLER FLIPS eval_parity before every such
       launch; the evaluator call that follows reads s->eval_parity from the struct it is handed, so it must be handed THE SAME
       tm_store value (parity included) as the tm_sim_step before it - not a stale copy.  tm_move_begin clears both sets.  A
       caller that does not flip lets a segment's counter grow from launch to launch: the append is bounded and the game is
       flagged TM_ERR_EVAL_LIST.  tm_sim_step / tm_move_begin return hipErrorInvalidValue when eval_list / eval_cnt are NULL. */
    int32_t *eval_list;   /* [G*eval_slots][2] (request slot, observation index) */
    int32_t *eval_cnt;    /* [G][2] entries per segment and parity (segment s of the games of *this at [s][parity]) */
    /* TM_KIND_VALUESIM / TM_KIND_CPPAGENT with TM_SIM_EVAL_NEEDED: the evaluator's output per observation, (v, var, epoch bits, 0);
       valid while epoch == eval_epoch (>= 1; the owner of the weights bumps it when they change); NULL: no cache */
    float *obs_eval;      /* [G][N][4] */
    int32_t eval_parity;  /* 0 / 1, see eval_list */
    int32_t eval_epoch;
    /* TM_KIND_DIST with online != 0: the distributions of the harvested nodes (DistValueSimOnline.store_nodes,
       agents/DistValueSimOnline.py:116-141: a freed node with >= min_visits_to_store visits whose seven children have all been
       visited; replay_obs = its packed observation, replay_stat = (0, 0, visits, 0)) */
    float *replay_dist;   /* [G][replay_cap][TM_DIST_ROW] */
    /* a launch over SOME of the games (the catch-up launches at the end of a move: a handful of games that spent launches
       collecting garbage still owe simulations): simulation wave i of tm_sim_step takes game game_list[i], i < n_listed; the
       collector workgroups look after all n_games as always.  NULL: every game (wave i = game i).  Filled by tm_sims_owing. */
    const int32_t *game_list;
    int32_t n_listed;
    int32_t gc_cost_units; /* cost units of bounded collection steps (init 1, count + re-insertion of the nodes 6, write + re-insertion of
                              the observations 7: about 5 microseconds a unit) the collector workgroups of one tm_sim_step launch take
                              on; 0: the default (13).  More units serve more collections per launch and make the launch longer. */
    int32_t gc_collectors; /* collector workgroups per tm_sim_step launch (half for the bounded steps, half for the marking); 0: the
                              default (128), at most 128.  A collection must be continued by launches with the same number. */
} tm_store;
#define TM_EVAL_SEGS(n_games) ((n_games) < 64 ? (n_games) : 64)

/* pools, free lists, tables, rng (seed 1), control blocks.  Everything else must be zero-filled by the caller. */
int tm_pool_init(const tm_store *s, void *stream);
/* re-initialise the trees of the games with mask[g] != 0 (pool exhausted beyond what GC can reclaim; the reference has
 * undefined behaviour there, agent.cpp:227-231); clears TM_ERR_POOL; the caller re-roots with tm_update_root */
int tm_pool_reset(const tm_store *s, const uint8_t *mask, void *stream);
/* host helper: fills host_table[n] = (float)norm_quantile((double)i) with this machine's libm */
void tm_fill_norm_quantile(float *host_table, int n);
void tm_fill_norm_quantile_f64(double *host_table, int n);      /* the same values in double (TM_KIND_DIST: nq_table_d) */

/* environment (batched pyTetris) */
int tm_env_init(const tm_store *s, const uint32_t *seeds, void *stream);                /* Tetris(...) */
int tm_env_step(const tm_store *s, const int32_t *actions, void *stream);               /* game.play(a) */
int tm_env_reset(const tm_store *s, const uint8_t *mask, void *stream);                 /* game.reset() where mask!=0 (NULL: where ended) */
int tm_env_render(const tm_store *s, int8_t *out /* [G][200] */, void *stream);         /* game.getState() */
int tm_env_info(const tm_store *s, int32_t *out /* [G][8]: end, score, lines, combo, ls[4] */, void *stream);

/* tree agent */
int tm_update_root(const tm_store *s, void *stream);                                    /* agent.update_root(game) */
/* TreeAgent's single calls as agents.cppmodule.agent exports them (agent.cpp:829 new_node, :833 expand, :828 remove_nodes;
 * agents/agent.py:90-145,246-257), one game state per tree.  games: device [G][16] packed games (the layout of
 * s->env_game; a pyTetris buffer in the reference); mask: device [G] or NULL (all games); out_idx: device [G] or NULL,
 * the node index new_node(game) returned (-1 where masked out).  tm_tree_expand = idx = new_node(game), then
 * child[idx][a] = new_node(game.play(a)) for the seven actions (agent.cpp:201-208), with the record's unique-child list.
 * A collection runs at the exhausting pop exactly as in the reference, so - as there - `game` must be reachable from the
 * root (or the pool must not run dry) for its node to survive the call. */
int tm_tree_new_node(const tm_store *s, const uint32_t *games, const uint8_t *mask, int32_t *out_idx, void *stream);
int tm_tree_expand(const tm_store *s, const uint32_t *games, const uint8_t *mask, int32_t *out_idx, void *stream);
int tm_tree_remove_nodes(const tm_store *s, const uint8_t *mask, void *stream);             /* agent.remove_nodes() */
/* One move of every game = tm_move_begin(sims), then launches of tm_sim_step(BACKUP|FRONT) each followed by the leaf
 * evaluator, until tm_sims_remaining reports 0: a launch starts a new simulation for a game only while the game has
 * quota left, backs up its pending one, or - when the game's node pool ran dry - runs one slice of its garbage
 * collection instead (s->gc_slice_cycles), so `sims` launches + 1 are enough unless a game collected in this move. */
int tm_move_begin(const tm_store *s, int sims, void *stream);
int tm_sims_remaining(const tm_store *s, int32_t *out /* device int[2]: max over games of launches still needed; games whose collection is under way */, void *stream);
/* the same, and WHICH games still need launches: out[2] = their number, list[0 .. out[2]) = their indices (any order; list:
 * device int32[n_games]) - what a caller puts into tm_store::game_list / n_listed for the launches that follow */
int tm_sims_owing(const tm_store *s, int32_t *out /* device int[3] */, int32_t *list, void *stream);
/* one step of every garbage collection under way (the collector workgroups of tm_sim_step alone, no simulation): what the
 * driver launches instead of whole simulation launches while it waits for collections at the end of a move */
int tm_gc_step(const tm_store *s, void *stream);
#define TM_SIM_BACKUP 1  /* finish the pending simulation: backup with eval_v/eval_var */
#define TM_SIM_FRONT 2   /* start one: select, expand, post evaluation requests into eval_obs */
#define TM_SIM_GC_FULL 4 /* a game that is collecting garbage finishes the collection in this launch (catch-up launches) */
#define TM_SIM_EVAL_NEEDED 8 /* the evaluator is a pure function of the observation (the built-in value net): post only the
                                requests whose outputs the backup will use - leaf-parallel kinds: the children on their first
                                visit (core.h:341-350; agent.cpp:536-545: and not finished); TM_KIND_VALUESIM / TM_KIND_CPPAGENT:
                                not a leaf whose observation was evaluated under the current weights (s->obs_eval).  Results
                                are identical either way; a Python evaluator callable keeps receiving all k children
                                (agent.cpp:424-436), so the Python-driven loop does not set it */
#define TM_SIM_GC_MEM_MARKS 16 /* the collector workgroups keep their mark bitmaps in memory whatever the pool's size (the form pools of
                                  more than 100 000 nodes take, tree.hip gc_marks_in_lds); set by tm_sim_step / tm_gc_step themselves when
                                  TM_GC_MARKS_IN_MEMORY=1 is in the environment (tests) */
int tm_sim_step(const tm_store *s, int flags, void *stream);
int tm_eval_render(const tm_store *s, int8_t *out /* [G*eval_slots][200] */, void *stream);

/* The store of games [first, first+n) of *s (every array is per game contiguous: a slice is the same struct with
 * offset pointers).  Sub-batches of one process, shards of a multi-GPU job.  `first` is a multiple of four (a tree-kernel
 * workgroup's games; the groups of TM_GS_GC_ACTIVE4): hipErrorInvalidValue otherwise. */
int tm_store_slice(const tm_store *s, int first, int n, tm_store *out);

/* Native driver of one move's search = the reference's `for i in range(sims)` loop (agents/ValueSim.py:76-94,
 * agents/agent.py:147-150, agent.cpp:407-460) as a host launch loop.  The games are cut into n_sub sub-batches on
 * their own HIP streams so that one sub-batch's tree kernel runs under another's value-net kernels; results per game do
 * not depend on n_sub.  tm_search_run returns when all `sims` simulations of every game are complete (it issues the
 * catch-up launches of games that collected garbage, tm_sims_remaining).  vn_params == NULL: no evaluator launches
 * (TM_KIND_VANILLA).  The evaluator follows the store's kind: the value net (tm_valuenet_forward_requests; vn_scratch:
 * n_games * eval_slots * TM_VALUENET_SCRATCH_MFMA floats, no initial contents required) or, for TM_KIND_DIST, the distributional head
 * (tm_distnet_forward_requests; vn_params / vn_prepared = its blobs, vn_scratch: n_games * TM_DISTNET_SCRATCH floats).  ev_every > 0: HIP events
 * around every ev_every-th simulation of sub-batch 0 (on the stream it runs on), read back by tm_search_stats:
 * out = {runs, tree launches, catch-up launches, timed samples, sum tree-kernel ms, sum value-net ms, n_sub}. */
typedef struct tm_search tm_search;
int tm_search_create(tm_search **out, const tm_store *s, int n_sub, int ev_every);
void tm_search_destroy(tm_search *h);
int tm_search_run(tm_search *h, int sims, const float *vn_params, const float *vn_prepared, float *vn_scratch,
                  void *stream);
int tm_search_stats(tm_search *h, double *out, int n, int reset);
/* the evaluator's weights changed: s->eval_epoch of the handle's copy of the store (outputs filed in obs_eval under another
 * epoch are not used; epochs are >= 1 and never reused for other weights) */
int tm_search_set_epoch(tm_search *h, int epoch);
int tm_root_stats(const tm_store *s, float *stats /* [G][3][7] */, int32_t *action /* [G] */, void *stream);
/* one game's tree in the reference's array layout (agents/agent.py:58-88), for inspection and tests */
int tm_export_game(const tm_store *s, int game, int32_t *child /* [N][7] */, float *score, int32_t *n_to_o,
                   int32_t *visit, float *value, float *variance, uint8_t *end_obs, void *stream);

/* agents.cppmodule.core (core.cpp:20-26) in batched form: B independent trees in the reference's own
 * array layout, tree b at offset b*n_nodes of every array.  rng: [B][32] as in tm_store. */
int tm_core_select_trace_obs(int n_trees, int n_nodes, const int32_t *roots, const int32_t *child,
                             const int32_t *visit, const float *value, const float *variance, const float *score,
                             const int32_t *n_to_o, int low, uint32_t *rng, const float *nq_table, int nq_size,
                             int32_t *trace /* [B][max_trace] */, int32_t *trace_len, int max_trace, void *stream);
int tm_core_backup_trace_obs(int n_trees, int n_nodes, const int32_t *trace, const int32_t *trace_len, int max_trace,
                             int32_t *visit, float *value, float *variance, const int32_t *n_to_o, const float *score,
                             const double *_value, const double *_variance, double gamma, void *stream);
int tm_core_backup_trace_obs_lp(int n_trees, int n_nodes, const int32_t *trace, const int32_t *trace_len,
                                int max_trace, int32_t *visit, float *value, float *variance, const int32_t *n_to_o,
                                const float *score, const uint8_t *end, const int32_t *_child /* [B][7] */,
                                const int32_t *_obs, const int32_t *k, const float *_value, const float *_variance,
                                double gamma, int mixture, int averaged, void *stream);
int tm_core_get_unique_child_obs(int n_trees, int n_nodes, const int32_t *index, const int32_t *child,
                                 const float *score, const int32_t *n_to_o, int32_t *c_nodes /* [B][7] */,
                                 int32_t *c_obs, int32_t *count, void *stream);
int tm_core_get_all_childs(int n_trees, int n_nodes, const int32_t *roots, const int32_t *child,
                           uint8_t *mark /* [B][n_nodes] */, int32_t *queue /* [B][n_nodes] scratch */, void *stream);

/* The array-level distribution helpers below (tm_dist_*, tm_distpy_*) and tm_core_get_all_childs are the compatibility
 * surface of the reference's function-per-call API: ONE LANE PER TREE / DISTRIBUTION (scalar code per lane, as the reference's
 * loops are written), meant for callers that keep the reference's arrays.  The engine's own search does this work at wave
 * level inside tm_sim_step (tree.hip: wave_dist_front / wave_dist_back, one lane per atom; the collectors' breadth-first
 * marking) and does not call them.
 *
 * distributional head helpers (agents/cppmodule/core.h:387-449; defined there, not exported by core.cpp): n categorical
 * distributions of `bins` atoms over [vmin, vmax).  tm_dist_transform: every source bin is an interval `scale` bins wide
 * shifted by shift[i] (in value units), its mass split over the two destination bins it overlaps (mass beyond the last
 * bin is dropped - the reference writes one float past its vector there).  tm_dist_mean_variance: out[i] = (mean, variance)
 * in double, bin centres vmin + (b + 1/2) delta accumulated as the reference does. */
int tm_dist_transform(int n, int bins, const float *dist /* [n][bins] */, double vmin, double vmax,
                      const double *shift /* [n] */, double scale, float *out /* [n][bins] */, void *stream);
int tm_dist_mean_variance(int n, int bins, const float *dist, double vmin, double vmax, double *out /* [n][2] */,
                          void *stream);

/* agents/core_distributional.py, the numba kernels of the reference's unfinished distributional agent, batched (one tree
 * or distribution per index): shift_distribution :12-37 (x[i] >= 0 in value units; mass reaching the top bin stays there),
 * policy_dist :66-79 (child_nodes[B][7] with n_child[B] valid entries, node_stats[B][n_nodes][5] = visit, mean, score,
 * variance, M2; returns the chosen child per tree), backup_trace_distributional :108-124 (updates node_stats and
 * node_dist[B][n_nodes][bins] along trace[B][max_trace]; scratch[B][bins]).  `fastmath` numba code: held to a float
 * tolerance (2e-6 relative), not to bit patterns.  Their mean_dist / mean_variance = tm_dist_mean_variance(vmin = 0,
 * vmax = vmax - vmin). */
int tm_distpy_shift(int n, int bins, const float *dist /* [n][bins] */, const double *x /* [n] */, double vmin, double vmax,
                    float *out /* [n][bins] */, void *stream);
int tm_distpy_policy(int n_trees, int n_nodes, const int32_t *child_nodes, const int32_t *n_child, const float *node_stats,
                     const double *curr_reward /* [B] */, int32_t *out /* [B] */, void *stream);
int tm_distpy_backup(int n_trees, int n_nodes, int bins, const int32_t *trace, const int32_t *trace_len, int max_trace,
                     float *node_stats, float *node_dist, const double *r /* [B] */, const float *leaf_dist /* [B][bins] */,
                     double vmin, double vmax, float *scratch, void *stream);

/* value network forward (model/model_vv.py:13-52; Model_VV.inference 210-217): states int8 [n][200] -> v[n], var[n].
 * params: 478342 floats in PyTorch state_dict layouts (order as in oracle/valuenet_oracle.c).
 * tm_valuenet_prepare re-lays the conv2/conv3/fc1 weights into MFMA operand streams (call after every weight
 * change); prepared: TM_VALUENET_PREPARED floats.  tm_valuenet_forward is the matrix-core path (scratch:
 * n x TM_VALUENET_SCRATCH_MFMA floats); tm_valuenet_forward_plain is the one-thread-per-output form with the
 * same fma-chain numerics (scratch: n x TM_VALUENET_SCRATCH floats).  Both are bit-identical by construction. */
/* The optimiser step of the online fit, model/yogi.py:39-90 (Yogi: lr, betas, eps, coupled weight decay) over FLAT device buffers:
 * p (parameters), g (gradients), m / v (exp_avg / exp_avg_sq), n floats each; state = 8 doubles on the device, zero before the first
 * step ([0] the step count, [1] [2] beta^t, [3] lr / (1 - beta1^t), [4] sqrt(1 - beta2^t): advanced on the device, so that the call is
 * the same two launches every time and can be captured in a HIP graph).  Per element the reference's operations in its order, fp32,
 * no contraction; step 1 sets exp_avg = 0, exp_avg_sq = g * g first (yogi.py:62-66).  Replaces the per-tensor loop of
 * Yogi.step (12 parameters x 8 launches). */
int tm_yogi_step(float *p, const float *g, float *m, float *v, double *state, int n, double lr, double beta1, double beta2,
                 double eps, double weight_decay, void *stream);
int tm_valuenet_prepare(const float *params, float *prepared, void *stream);
int tm_valuenet_forward(const float *params, const float *prepared, const int8_t *states, int n, float *v, float *var,
                        float *scratch, void *stream);
/* evaluate the tree engine's pending requests (s->eval_obs) and write s->eval_v / s->eval_var; the packed
 * observations are rendered inside the first kernel (no int8 staging).  scratch: G*eval_slots x TM_VALUENET_SCRATCH_MFMA */
int tm_valuenet_forward_requests(const float *params, const float *prepared, const tm_store *s, float *scratch,
                                 void *stream);
int tm_valuenet_forward_plain(const float *params, const int8_t *states, int n, float *v, float *var, float *scratch,
                              void *stream);

/* distributional value head (model/model_distributional.py:18-57 `Net`, Model_Dist.inference :100-107), the leaf evaluator
 * of TM_KIND_DIST: states int8 [n][200] (the 20 visible rows; the net's two extra rows on top are empty) -> softmax over
 * `atoms` (<= 64) bins, dist[i * dist_stride + b].  params: TM_DISTNET_PARAMS(atoms) floats in PyTorch state_dict order and
 * layouts; tm_distnet_prepare re-lays conv2 / fc1 into MFMA operand streams (after every weight change).  fp32 matrix cores,
 * one k-ordered fma chain per pre-activation (oracle/distnet_oracle.c computes the same bits).  scratch: n x
 * TM_DISTNET_SCRATCH floats, no initial contents required. */
int tm_distnet_prepare(const float *params, float *prepared, void *stream);
int tm_distnet_forward(const float *params, const float *prepared, const int8_t *states, int n, int atoms, float *dist,
                       int dist_stride, float *scratch, void *stream);
/* the tree engine's pending requests (TM_KIND_DIST: s->eval_obs[g] = the leaf NODE of game g, rendered from its packed game
 * inside the first kernel) -> s->eval_dist[g][0 .. dist_bins); scratch: n_games x TM_DISTNET_SCRATCH floats */
int tm_distnet_forward_requests(const float *params, const float *prepared, const tm_store *s, float *scratch, void *stream);

const char *tm_version(void);
/* sizeof(tm_store) and a few offsets, so a host mirror of the struct can be checked without a GPU */
int tm_store_layout(int *out, int n);
#ifdef __cplusplus
}
#endif
#endif
