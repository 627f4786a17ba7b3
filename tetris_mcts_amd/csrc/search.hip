// Host-side driver of one move's search over all games of this process: the loop the reference runs in Python
// (TreeAgent.play -> mcts(root, sims), agents/agent.py:147-150; ValueSim.mcts, agents/ValueSim.py:76-94;
// ValueSimLP.mcts, agents/ValueSimLP.py:44-70; MCTSAgent::play, agents/cppmodule/agent.cpp:407-460), as a native
// launch loop.
//
// The games are cut into `n_sub` contiguous sub-batches, each with its own HIP stream.  A simulation of one game is a
// strict chain  tree kernel (backup k-1, select + expand k) -> value net (leaf of k) -> tree kernel ...,  but the
// chains of different sub-batches are independent, so with n_sub > 1 the tree walk of one sub-batch (latency-bound
// pointer chasing, a few waves per SIMD, no matrix work) runs under the value-net kernels of another (MFMA-bound).
// Per-game results do not depend on n_sub (every game's sequence of events is its own).
//
// A game whose node pool runs dry does not simulate while the collector workgroups of the following launches collect its
// garbage (tree.hip gc_collector_block), so it falls behind its quota; after the `sims` regular launches the driver
// finishes the collections still under way with collector-only launches (tm_gc_step), asks how many launches are still
// owed (tm_sims_remaining) and issues them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/tetris_mcts_hip.h"

struct tm_search {
    tm_store full;
    int n_sub;
    bool own_streams;
    std::vector<tm_store> sub;
    std::vector<int> first;
    std::vector<hipStream_t> streams;
    hipEvent_t ev_start = nullptr;
    std::vector<hipEvent_t> ev_done;
    int32_t* rem_dev = nullptr;      // per sub-batch: launches still needed, collections under way, games that owe launches
    int32_t* rem_host = nullptr;
    int32_t* owing_dev = nullptr;    // [n_games] the games that owe launches, per sub-batch at its first game (tm_sims_owing)
    long long extra_waves = 0;       // simulation waves of the catch-up launches (a full launch has n_games of them)
    // HIP events around the regular launches of a move (sub-batch 0's stream): the time the per-kernel figures must add up to
    hipEvent_t ev_loop0 = nullptr, ev_loop1 = nullptr;
    double loop_ms = 0;
    long long loop_sims = 0;
    // HIP-event timing of every `ev_every`-th simulation of sub-batch 0, on the stream the kernels run on
    int ev_every;
    std::vector<hipEvent_t> ev;      // triples: before value net, after value net (= before tree), after tree
    int ev_used;
    double tree_ms, nn_ms;
    long long n_timed, n_runs, extra_launches, launches, gc_launches;
};

#define TM_TRY(x) do { int e_ = (int)(x); if (e_ != 0) return e_; } while (0)

extern "C" {

int tm_store_slice(const tm_store* s, int first, int n, tm_store* out) {
    // (a slice begins at a multiple of four games: a tree-kernel workgroup's games, and the groups of TM_GS_GC_ACTIVE4)
    if (first < 0 || (first & 3) || n < 0 || first + n > s->n_games) return (int)hipErrorInvalidValue;
    tm_store t = *s;
    const size_t f = (size_t)first, N = (size_t)s->max_nodes;
    const size_t bm_bytes = ((N + 7) / 8 + 15) & ~(size_t)15;
    t.n_games = n;
    t.node_rec += f * N * TM_REC_DW;
    t.node_child += f * N * TM_KIDS_DW;
    t.node_game += f * N * TM_GAME_DW;
    t.obs_stat += f * N * 4;
    t.obs_key += f * N * TM_OBS_DW;
    t.node_tab += f * (size_t)s->table_cap;
    t.obs_tab += f * (size_t)s->table_cap;
    t.free_node += f * N;
    t.free_obs += f * N;
    t.gs += f * TM_GS_DW;
    t.rng += f * 32;
    if (t.env_game) t.env_game += f * TM_GAME_DW;
    if (t.env_line_stats) t.env_line_stats += f * 4;
    t.trace += f * (size_t)s->max_trace * 4;
    t.leaf += f * TM_LEAF_DW;
    t.eval_obs += f * (size_t)s->eval_slots;
    t.eval_v += f * (size_t)s->eval_slots;
    t.eval_var += f * (size_t)s->eval_slots;
    t.gc_mark += f * 2 * bm_bytes;
    t.gc_queue += f * N;
    t.gc_part += f * TM_GC_PART_DW;
    if (s->kind == TM_KIND_DIST) { t.node_dist += f * N * TM_DIST_ROW; t.eval_dist += f * TM_DIST_ROW; }
    if (t.replay_obs) t.replay_obs += f * (size_t)s->replay_cap * TM_OBS_DW;
    if (t.replay_stat) t.replay_stat += f * (size_t)s->replay_cap * 4;
    if (t.replay_count) t.replay_count += f;
    if (t.mt_state && (s->kind == TM_KIND_VANILLA || s->kind == TM_KIND_VANILLA_C)) t.mt_state += f * 625;
    t.eval_list += f * (size_t)s->eval_slots * 2;
    t.eval_cnt += f * 2;
    if (t.obs_eval) t.obs_eval += f * N * 4;
    if (t.replay_dist) t.replay_dist += f * (size_t)s->replay_cap * TM_DIST_ROW;
    *out = t;
    return 0;
}

void tm_search_destroy(tm_search* h);

int tm_search_create(tm_search** out, const tm_store* s, int n_sub, int ev_every) {
    if (n_sub < 1 || n_sub > 64 || n_sub > s->n_games) return (int)hipErrorInvalidValue;
    tm_search* h = new tm_search();
    h->full = *s;
    h->n_sub = n_sub;
    h->own_streams = n_sub > 1;
    h->ev_every = ev_every;
    h->ev_used = 0;
    h->tree_ms = h->nn_ms = 0;
    h->n_timed = h->n_runs = h->extra_launches = h->launches = h->gc_launches = 0;
    // every failure leaves through the same door: whatever exists by then is released by tm_search_destroy
    auto fail = [&](int e) { tm_search_destroy(h); return e; };
    // sub-batch boundaries on multiples of 4 games (one workgroup of the tree kernel = 4 games)
    const int G = s->n_games;
    int per = ((G + n_sub - 1) / n_sub + 3) & ~3;
    for (int k = 0, f = 0; k < n_sub; ++k) {
        int n = (k == n_sub - 1) ? G - f : (per < G - f ? per : G - f);
        if (n < 0) n = 0;
        tm_store t;
        int e = tm_store_slice(s, f, n, &t);
        if (e) return fail(e);
        h->sub.push_back(t);
        h->first.push_back(f);
        f += n;
    }
    hipError_t e = hipEventCreateWithFlags(&h->ev_start, hipEventDisableTiming);
    if (e != hipSuccess) return fail((int)e);
    e = hipEventCreate(&h->ev_loop0);
    if (e != hipSuccess) return fail((int)e);
    e = hipEventCreate(&h->ev_loop1);
    if (e != hipSuccess) return fail((int)e);
    for (int k = 0; k < n_sub; ++k) {
        hipStream_t st = nullptr;
        if (h->own_streams) {
            e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            if (e != hipSuccess) return fail((int)e);
        }
        h->streams.push_back(st);
        hipEvent_t d;
        e = hipEventCreateWithFlags(&d, hipEventDisableTiming);
        if (e != hipSuccess) return fail((int)e);
        h->ev_done.push_back(d);
    }
    e = hipMalloc(&h->rem_dev, sizeof(int32_t) * 3 * n_sub);
    if (e != hipSuccess) return fail((int)e);
    e = hipHostMalloc(&h->rem_host, sizeof(int32_t) * 3 * n_sub, hipHostMallocDefault);
    if (e != hipSuccess) return fail((int)e);
    e = hipMalloc(&h->owing_dev, sizeof(int32_t) * (size_t)(G > 0 ? G : 1));
    if (e != hipSuccess) return fail((int)e);
    *out = h;
    return 0;
}

void tm_search_destroy(tm_search* h) {
    if (!h) return;
    for (auto st : h->streams) if (st && h->own_streams) (void)hipStreamDestroy(st);
    for (auto e : h->ev_done) (void)hipEventDestroy(e);
    for (auto e : h->ev) (void)hipEventDestroy(e);
    if (h->ev_start) (void)hipEventDestroy(h->ev_start);
    if (h->ev_loop0) (void)hipEventDestroy(h->ev_loop0);
    if (h->ev_loop1) (void)hipEventDestroy(h->ev_loop1);
    if (h->rem_dev) (void)hipFree(h->rem_dev);
    if (h->owing_dev) (void)hipFree(h->owing_dev);
    if (h->rem_host) (void)hipHostFree(h->rem_host);
    delete h;
}

// One move's search: `sims` simulations for every game.  vn_params == NULL: no leaf evaluator launch (TM_KIND_VANILLA).
// Synchronises with the host (it has to read how many catch-up launches are needed); on return all work is complete.
int tm_search_run(tm_search* h, int sims, const float* vn_params, const float* vn_prepared, float* vn_scratch,
                  void* stream_) {
    hipStream_t caller = (hipStream_t)stream_;
    const int K = h->n_sub;
    TM_TRY(tm_move_begin(&h->full, sims, caller));
    std::vector<hipStream_t> st(K);
    for (int k = 0; k < K; ++k) st[k] = h->own_streams ? h->streams[k] : caller;
    if (h->own_streams) {
        TM_TRY(hipEventRecord(h->ev_start, caller));
        for (int k = 0; k < K; ++k) TM_TRY(hipStreamWaitEvent(st[k], h->ev_start, 0));
    }
    // `owing` > 0: a launch over the games of tm_sims_owing's list only (tm_store::game_list)
    auto step = [&](int k, int owing = 0) -> int {
        if (h->sub[k].n_games == 0) return 0;
        h->launches += 1;
        h->sub[k].eval_parity ^= 1;      // the dense request list: every launch appends under the other parity (tm_store::eval_list)
        h->sub[k].game_list = owing > 0 ? h->owing_dev + h->first[k] : nullptr;
        h->sub[k].n_listed = owing;
        // the evaluator is the built-in net, a function of the observation: only the requests whose outputs the backup uses
        const int e = tm_sim_step(&h->sub[k], TM_SIM_BACKUP | TM_SIM_FRONT | (vn_params ? TM_SIM_EVAL_NEEDED : 0), st[k]);
        h->sub[k].game_list = nullptr;
        h->sub[k].n_listed = 0;
        return e;
    };
    auto nn = [&](int k) -> int {
        if (!vn_params || h->sub[k].n_games == 0) return 0;
        if (h->full.kind == TM_KIND_DIST)      // the distributional head (distnet.hip): eval_obs names the leaf node
            return tm_distnet_forward_requests(vn_params, vn_prepared, &h->sub[k],
                                               vn_scratch + (size_t)h->first[k] * TM_DISTNET_SCRATCH, st[k]);
        float* scr = vn_scratch + (size_t)h->first[k] * h->full.eval_slots * TM_VALUENET_SCRATCH_MFMA;
        return tm_valuenet_forward_requests(vn_params, vn_prepared, &h->sub[k], scr, st[k]);
    };
    h->ev_used = 0;
    TM_TRY(hipEventRecord(h->ev_loop0, st[0]));
    for (int k = 0; k < K; ++k) TM_TRY(step(k));
    for (int i = 0; i < sims; ++i) {
        const bool timed = h->ev_every > 0 && (i % h->ev_every) == 0;
        for (int k = 0; k < K; ++k) {
            hipEvent_t* e3 = nullptr;
            if (timed && k == 0) {
                while ((int)h->ev.size() < h->ev_used + 3) {
                    hipEvent_t e;
                    TM_TRY(hipEventCreate(&e));
                    h->ev.push_back(e);
                }
                e3 = &h->ev[h->ev_used];
                h->ev_used += 3;
                TM_TRY(hipEventRecord(e3[0], st[k]));
            }
            TM_TRY(nn(k));
            if (e3) TM_TRY(hipEventRecord(e3[1], st[k]));
            TM_TRY(step(k));
            if (e3) TM_TRY(hipEventRecord(e3[2], st[k]));
        }
    }
    TM_TRY(hipEventRecord(h->ev_loop1, st[0]));
    // catch-up: games that spent launches collecting garbage still owe simulations.  Collections still under way are
    // finished first, by collector-only launches (tm_gc_step: a step of every collection each, no simulation, no evaluator).
    // The catch-up launches run over the games that owe only (a handful of 4096: tm_sims_owing lists them, tm_store::game_list),
    // so their tree kernel is a few workgroups beside the collectors and the evaluator draws a few requests from the dense list.
    for (long long rounds = 0;; ++rounds) {
        // (every round finishes simulations or steps of collections; a move that needs this many is not making progress)
        if (rounds > 64LL * sims + 100000) return (int)hipErrorLaunchFailure;
        for (int k = 0; k < K; ++k) {
            h->rem_host[3 * k] = h->rem_host[3 * k + 1] = h->rem_host[3 * k + 2] = 0;
            if (h->sub[k].n_games == 0) continue;
            TM_TRY(tm_sims_owing(&h->sub[k], h->rem_dev + 3 * k, h->owing_dev + h->first[k], st[k]));
            TM_TRY(hipMemcpyAsync(h->rem_host + 3 * k, h->rem_dev + 3 * k, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st[k]));
        }
        for (int k = 0; k < K; ++k) TM_TRY(hipStreamSynchronize(st[k]));
        int r = 0, collecting = 0;
        for (int k = 0; k < K; ++k) { r = h->rem_host[3 * k] > r ? h->rem_host[3 * k] : r; collecting += h->rem_host[3 * k + 1]; }
        if (r == 0) break;
        if (collecting > 0) {
            for (int i = 0; i < 6; ++i)
                for (int k = 0; k < K; ++k)
                    if (h->rem_host[3 * k + 1] > 0) { TM_TRY(tm_gc_step(&h->sub[k], st[k])); h->gc_launches += 1; }
            continue;
        }
        for (int i = 0; i < r; ++i)
            for (int k = 0; k < K; ++k)
                if (h->rem_host[3 * k] > i) {
                    TM_TRY(nn(k));
                    TM_TRY(step(k, h->rem_host[3 * k + 2]));
                    h->extra_launches += 1;
                    h->extra_waves += h->rem_host[3 * k + 2];
                }
    }
    for (int j = 0; j + 2 < h->ev_used; j += 3) {
        float a = 0, b = 0;
        if (hipEventElapsedTime(&a, h->ev[j], h->ev[j + 1]) == hipSuccess &&
            hipEventElapsedTime(&b, h->ev[j + 1], h->ev[j + 2]) == hipSuccess) {
            h->nn_ms += a;
            h->tree_ms += b;
            h->n_timed += 1;
        }
    }
    {
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->ev_loop0, h->ev_loop1) == hipSuccess) { h->loop_ms += ms; h->loop_sims += sims; }
    }
    h->n_runs += 1;
    return 0;
}

// The weights of the built-in evaluator changed (or are another model's): outputs filed under another epoch are not used
// (tm_store::obs_eval, TM_SIM_EVAL_NEEDED).  The handle holds its own copy of the store description.
int tm_search_set_epoch(tm_search* h, int epoch) {
    h->full.eval_epoch = epoch;
    for (auto& t : h->sub) t.eval_epoch = epoch;
    return 0;
}

// out[0..10] = runs, tree-kernel launches, catch-up launches, timed samples, sum of tree-kernel ms, sum of value-net ms,
// sub-batches, collector-only launches, sum of the regular launch loops' ms (HIP events around sims x (value net, tree
// kernel) of sub-batch 0's stream), simulations in those loops, simulation waves of the catch-up launches (each runs over the
// games that owe only); the per-kernel sums are over the timed samples (sub-batch 0, every ev_every-th simulation)
int tm_search_stats(tm_search* h, double* out, int n, int reset) {
    double v[11] = {(double)h->n_runs, (double)h->launches, (double)h->extra_launches, (double)h->n_timed, h->tree_ms,
                    h->nn_ms, (double)h->n_sub, (double)h->gc_launches, h->loop_ms, (double)h->loop_sims, (double)h->extra_waves};
    for (int i = 0; i < n && i < 11; ++i) out[i] = v[i];
    if (reset) {
        h->tree_ms = h->nn_ms = h->loop_ms = 0;
        h->n_timed = h->n_runs = h->extra_launches = h->launches = h->gc_launches = h->loop_sims = h->extra_waves = 0;
    }
    return 11;
}

}  // extern "C"
