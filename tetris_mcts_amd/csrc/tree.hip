// Batched Tetris-MCTS tree engine for gfx950: one wavefront per game.
//
// Reference behaviour reproduced (bit for bit on integers and floats):
//   new_node / expand      agents/agent.py:90-145 (agent.cpp:201-264)
//   select_trace_obs       agents/cppmodule/core.h:167-224 (check_low 65-77, policy_clt 83-105,
//                          get_unique_child_obs 111-144)
//   backup_trace_obs       core.h:226-260 ; backup_trace_obs_LP core.h:303-381 ;
//   MCTSAgent twins        agent.cpp:496-566
//   compute_stats/get_action agents/agent.py:153-185
// Layout and kernel design: DESIGN.md.  Node record (96 B) = the node's unique children in selection order as
// (child, observation, score) triples + (header, own observation, own score): get_unique_child_obs is evaluated
// once, at expansion, and a tree level costs one memory round trip (children's statistics + speculative prefetch
// of the children's records).  The raw per-action child indices live in a separate 32-byte row (GC, root statistics).
#include <hip/hip_runtime.h>
#include <math.h>
#include <atomic>
#include <stdlib.h>
#include "engine.h"
#include "../../include/tetris_mcts_hip.h"

namespace tmcts {

constexpr int WPB = 4;  // wavefronts (games) per workgroup
constexpr int TRACE_LDS = 64;   // trace entries buffered in LDS before a coalesced flush
struct WaveLds {
    uint32_t slots[8][GAME_DW];  // 0..6 successor games, 7 the parent
    uint32_t okeys[7][OBS_DW];
    uint32_t misc[64];
    uint4 tbuf[TRACE_LDS];       // piece 7 (predicted child, observation, score bits, header) of the nodes of the walk in progress
    uint4 tdummy[64];            // where the lanes that do not own the trace put their (ignored) copy: no exec masking in the walk
};
struct MtLds { uint32_t mt[624]; uint32_t idx; uint32_t pad[3]; };   // CPython random state (TM_KIND_VANILLA)

// CPython's genrand_uint32 + random.randint(0, 6) (= _randbelow_with_getrandbits(7): 3 bits, rejection), one lane
__device__ inline uint32_t mt_next(MtLds& m) {
    if (m.idx >= 624) {
        for (int kk = 0; kk < 624; ++kk) {
            uint32_t y = (m.mt[kk] & 0x80000000u) | (m.mt[(kk + 1) % 624] & 0x7fffffffu);
            m.mt[kk] = m.mt[(kk + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        m.idx = 0;
    }
    uint32_t y = m.mt[m.idx];
    m.idx += 1;
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
__device__ inline int mt_randint7(MtLds& m) {
    uint32_t r = mt_next(m) >> 29;
    while (r >= 7) r = mt_next(m) >> 29;
    return (int)r;
}
// random.randint(0, 7) (agents/VanillaC.py:7) = _randbelow_with_getrandbits(8): 8.bit_length() = 4 bits, rejection of r >= 8
__device__ inline int mt_randint8(MtLds& m) {
    uint32_t r = mt_next(m) >> 28;
    while (r >= 8) r = mt_next(m) >> 28;
    return (int)r;
}

// Base pointers of one game, computed where they are used (g is wave-uniform, so each is a few scalar instructions):
// holding all fifteen of them in scalar registers for the whole kernel made the tree walk's loop reload spilled ones.
struct GP {
    const tm_store& S;
    int g;
    __device__ __forceinline__ size_t n() const { return (size_t)S.max_nodes; }
    __device__ __forceinline__ uint32_t* rec() const { return S.node_rec + (size_t)g * n() * TM_REC_DW; }
    __device__ __forceinline__ uint32_t* kids() const { return S.node_child + (size_t)g * n() * TM_KIDS_DW; }
    __device__ __forceinline__ uint32_t* game() const { return S.node_game + (size_t)g * n() * TM_GAME_DW; }
    __device__ __forceinline__ uint32_t* stat() const { return S.obs_stat + (size_t)g * n() * 4; }
    __device__ __forceinline__ uint32_t* okey() const { return S.obs_key + (size_t)g * n() * TM_OBS_DW; }
    __device__ __forceinline__ uint64_t* ntab() const { return S.node_tab + (size_t)g * (size_t)S.table_cap; }
    __device__ __forceinline__ uint64_t* otab() const { return S.obs_tab + (size_t)g * (size_t)S.table_cap; }
    __device__ __forceinline__ int32_t* fnode() const { return S.free_node + (size_t)g * n(); }
    __device__ __forceinline__ int32_t* fobs() const { return S.free_obs + (size_t)g * n(); }
    __device__ __forceinline__ int32_t* gs() const { return S.gs + (size_t)g * TM_GS_DW; }
    __device__ __forceinline__ int32_t* leaf() const { return S.leaf + (size_t)g * TM_LEAF_DW; }
    __device__ __forceinline__ int32_t* eval_obs() const { return S.eval_obs + (size_t)g * S.eval_slots; }
    __device__ __forceinline__ float* eval_v() const { return S.eval_v + (size_t)g * S.eval_slots; }
    __device__ __forceinline__ float* eval_var() const { return S.eval_var + (size_t)g * S.eval_slots; }
    __device__ __forceinline__ uint32_t* trace() const { return S.trace + (size_t)g * (size_t)S.max_trace * 4; }
};
// The summary of the phase words (TM_GS_GC_ACTIVE4: a bit a game in the control block of every fourth game): what lets a
// launch's collector workgroups find the collecting games without reading every game's control block - 4 096 scattered lines a
// workgroup, 128 workgroups, every launch: 1.7 us of every k_sim_step launch, measured by leaving the look out (r06).  Set by
// whoever makes a phase word non-zero, cleared by whoever owns the game when the word goes back to zero.
__device__ __forceinline__ void gc_active_set(const GP& P) { atomicOr(&P.S.gs[(size_t)(P.g & ~3) * TM_GS_DW + TM_GS_GC_ACTIVE4], 1 << (P.g & 3)); }
__device__ __forceinline__ void gc_active_clear(const GP& P) { atomicAnd(&P.S.gs[(size_t)(P.g & ~3) * TM_GS_DW + TM_GS_GC_ACTIVE4], ~(1 << (P.g & 3))); }

__device__ __forceinline__ GP game_ptrs(const tm_store& S, int g) { return GP{S, g}; }

__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ uint32_t shfl_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
// wave-uniform source lane: v_readlane_b32 (scalar path, a few cycles) instead of an LDS-crossbar ds_bpermute
__device__ __forceinline__ uint32_t rl_u32(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
#define GSV(snap, word) ((int)rl_u32((uint32_t)(snap), (word)))     /* word `word` of a control-block snapshot */
__device__ __forceinline__ float rl_f32(float v, int src) { return __uint_as_float(rl_u32(__float_as_uint(v), src)); }
// DPP exchanges inside an 8-lane group (no LDS, no scalar round trip): partner = lane^1, lane^2, 7-lane
__device__ __forceinline__ uint32_t dpp_x1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t dpp_x2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t dpp_hm(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true); }
// 16-byte load at a 32-bit byte offset from a wave-uniform base (global_load_dwordx4 v, v_off, s[base:base+1]): the per-game
// arrays are < 4 GiB, so the walk needs one shift-add per address instead of 64-bit pointer arithmetic
__device__ __forceinline__ uint4 ld16(const uint32_t* base, uint32_t byte_off) {
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + byte_off);
}
// The same through a buffer resource (base, size) in four scalar registers: buffer_load_dwordx4 v, v_off, s[rsrc], 0 offen.
// An intrinsic call is one indivisible 16-byte load for the optimizer (a plain uint4 load was split, and the part only
// one branch needs was sunk into that branch = a second memory round trip per tree level).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
    // wave-uniform by construction (one game per wave): pin base and size to scalar registers, or every load through the
    // resource becomes a readfirstlane waterfall loop
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    const int nb = __builtin_amdgcn_readfirstlane((int)(uint32_t)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
}
__device__ __forceinline__ uint4 buf_ld16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t bperm_u32(int byte_addr, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v);
}
// Wave-wide reductions of a value that is replicated inside each 8-lane group (one group per unique child): only the
// strides 8, 16 and 32 are needed = row_shr:8, row_bcast:15 (rows 1, 3), row_bcast:31 (rows 2, 3); the result is in lane 63.
__device__ __forceinline__ uint32_t group_sum_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return rl_u32(v, 63);
}
__device__ __forceinline__ int group_max_i32(int v) {
    // v_max_i32 with a DPP source: lanes whose source lane is invalid or whose row is masked keep their value.  One
    // instruction per stride (the builtin form costs a move, a DPP move and a max); 2 wait states between a VALU write
    // and a DPP read of the same register.
    asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0" : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    uint32_t lo = shfl_u32((uint32_t)v, src), hi = shfl_u32((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t rl_u64(uint64_t v, int src) {   // wave-uniform source lane
    return ((uint64_t)rl_u32((uint32_t)(v >> 32), src) << 32) | rl_u32((uint32_t)v, src);
}

// ---------------------------------------------------------------------------------------------------
// glibc rand() (TYPE_3, r[i] = r[i-3] + r[i-31]; core.h:62,76 call the process-global generator, here
// one stream per game, all lanes of the wave step it redundantly on the LDS copy)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_rand(uint32_t& rs, int& pos) {
    int f = pos & 0xFF, b = (pos >> 8) & 0xFF;
    uint32_t val = rl_u32(rs, f) + rl_u32(rs, b);
    rs = ((int)(threadIdx.x & 63) == f) ? val : rs;
    f += 1;
    if (f >= 31) { f = 0; b += 1; }
    else { b += 1; if (b >= 31) b = 0; }
    pos = f | (b << 8);
    return val >> 1;
}

// the same on a copy of the state in LDS (L.misc[0..31]): the walk keeps no loop-carried vector register for it
__device__ __forceinline__ uint32_t wave_rand_lds(uint32_t* r, int& pos, int lane) {
    int f = pos & 0xFF, b = (pos >> 8) & 0xFF;
    const uint32_t val = r[f] + r[b];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) r[f] = val;
    __builtin_amdgcn_wave_barrier();
    f += 1;
    if (f >= 31) { f = 0; b += 1; }
    else { b += 1; if (b >= 31) b = 0; }
    pos = f | (b << 8);
    return val >> 1;
}

__device__ inline void srand_state(uint32_t* r /* 32 words */, int& pos, uint32_t seed) {
    if (seed == 0) seed = 1;
    int32_t word = (int32_t)seed;
    r[0] = (uint32_t)word;
    for (int i = 1; i < 31; ++i) {
        long long hi = word / 127773, lo = word % 127773;
        long long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int32_t)w;
        r[i] = (uint32_t)word;
    }
    r[31] = 0;
    int f = 3, b = 0;
    for (int i = 0; i < 310; ++i) {
        r[f] = r[f] + r[b];
        f += 1;
        if (f >= 31) { f = 0; b += 1; }
        else { b += 1; if (b >= 31) b = 0; }
    }
    pos = f | (b << 8);
}

// special.h:26-33, used only beyond the host-built table (counted in TM_GS_N_NQ_FALLBACK)
__device__ __attribute__((noinline)) float norm_quantile_dev(double t) {
    double alpha = 1 - 1 / t;
    return (float)(10 * log(1 - log(-log(alpha) / log(2.0)) / log(22.0)) / log(41.0));
}

// ---------------------------------------------------------------------------------------------------
// new_node for up to 7 candidate games held in L.slots[0..n) (agents/agent.py:90-130 applied in action
// order).  Lane a (< n) owns candidate a.  Returns the node and observation index per lane.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool eq_lds(const uint32_t* a, const uint32_t* b, int ndw) {
    uint32_t d = 0;
    for (int i = 0; i < ndw; ++i) d |= a[i] ^ b[i];
    return d == 0;
}
__device__ __forceinline__ bool eq_glb(const uint32_t* lds, const uint32_t* glb, int nq /* uint4 count */) {
    const uint4* q = reinterpret_cast<const uint4*>(glb);
    uint32_t d = 0;
    for (int i = 0; i < nq; ++i) {
        uint4 v = q[i];
        d |= (v.x ^ lds[4 * i]) | (v.y ^ lds[4 * i + 1]) | (v.z ^ lds[4 * i + 2]) | (v.w ^ lds[4 * i + 3]);
    }
    return d == 0;
}

// Probe an open-addressing table (slot = tag<<32 | index, index 0 = empty).  Returns the index of
// the entry whose stored record equals `key`, or 0 with `ins` = first empty slot.
__device__ __forceinline__ int table_find(const uint64_t* tab, uint32_t mask, uint64_t h, const uint32_t* key,
                                          const uint32_t* pool, int rec_dw, uint32_t& ins, bool& full) {
    uint32_t tag = (uint32_t)(h >> 32);
    uint32_t s = (uint32_t)h & mask;
    for (uint32_t probes = 0; probes <= mask; ++probes) {
        uint64_t e = tab[s];
        uint32_t idx = (uint32_t)e;
        if (idx == 0) { ins = s; return 0; }
        if ((uint32_t)(e >> 32) == tag && eq_glb(key, pool + (size_t)idx * rec_dw, rec_dw / 4)) return (int)idx;
        s = (s + 1) & mask;
    }
    full = true;
    ins = 0;
    return 0;
}

// Insert entries of the lanes in `need` one lane at a time (ascending lane = action order), skipping
// slots claimed earlier in this call (another lane of this wave may have probed the same empty slot).
__device__ __forceinline__ void table_insert_seq(uint64_t* tab, uint32_t mask, uint64_t need, int n, int lane,
                                                 uint64_t h, uint32_t ins, int idx, uint32_t* claimed) {
    int ncl = 0;
    for (int b = 0; b < n; ++b) {
        if (!((need >> b) & 1ull)) continue;
        if (lane == b) {
            uint32_t s = ins;
            for (;;) {
                bool cl = false;
                for (int j = 0; j < ncl; ++j) cl |= (claimed[j] == s);
                if (!cl && (s == ins || (uint32_t)tab[s] == 0)) break;
                s = (s + 1) & mask;
            }
            tab[s] = ((h >> 32) << 32) | (uint32_t)idx;
            claimed[ncl] = s;
        }
        ncl += 1;
        wave_sync();
    }
}

__device__ __forceinline__ void gc_wave(const tm_store& S, const GP& P, int g, int lane);

// The collectors' marker (gc_sweep_mark below) works through the index range in BLOCKS of GC_BLOCK_NODES nodes - GC_ROWS child
// rows per lane of ONE WAVE - and keeps the flags the marking workgroups of a game exchange, 128 of them in the game's control
// block, per CHUNK of the index range (GC_CHUNK_MIN_LOG2 nodes or more: a power of two of blocks, a node's chunk is a shift).
constexpr int GC_ROWS = 4;
constexpr int GC_BLOCK_LOG2 = 8;
constexpr int GC_BLOCK_NODES = GC_ROWS * 64;
static_assert((1 << GC_BLOCK_LOG2) == GC_BLOCK_NODES, "a block is 256 nodes");
constexpr int GC_CHUNK_MIN_LOG2 = 10;
__host__ __device__ inline int gc_chunk_log2(int N) {
    int cl = GC_CHUNK_MIN_LOG2;
    while (((long long)(32 * TM_GC_FLAG_WORDS) << cl) < N) ++cl;
    return cl;
}
__host__ __device__ inline int gc_chunk_nodes(int N) { return 1 << gc_chunk_log2(N); }
__host__ __device__ inline size_t gc_bm_bytes(int N) { return (((size_t)N + 7) / 8 + 15) & ~(size_t)15; }

constexpr int GC_REQ = 1, GC_DONE = 7;   // phase word: (launch << 4) | GC_REQ requested, 2..6 under way (GCP_*), (launch << 4) | GC_DONE complete
// SPECULATIVE MARKING: a game that is about to run out of nodes (S.gc_spec_nodes left) has its tree marked while it goes
// on simulating - the marking is what keeps a collecting game out for twenty launches.  Marks only ever grow while the
// root stays (nothing becomes unreachable inside a move), and the expansion keeps the marker's invariant ("what is
// reachable is marked or will be reached") with a write barrier: every successor it links is marked (a new node, with
// its observation) or, if it is an existing node found by transposition and not marked yet, marked and appended to the
// marker's queue.  At the exhausting pop the marking is simply continued to its end (usually a launch or two) and the
// collection proceeds; update_root drops a speculative marking (the root moves, marks would be stale).
//   (launch << 4) | GC_SPEC_REQ   asked for by the game's wave; collectors of LATER launches initialise the bitmaps
//   GC_SPEC_MARK                  marking under way, the game simulates with the barrier
//   (launch << 4) | GC_REQ_SPEC   the pool ran dry during GC_SPEC_MARK: clear the tables, finish the marking, go on
//   (launch << 4) | GC_REQ_OVER   the pool ran dry while GC_SPEC_REQ was pending: an ordinary request
// A phase word that a game's wave changes in the middle of a launch must read the same to every collector workgroup of
// that launch, whichever value it happens to load (they all have to agree on the launch's plan): the word the wave
// writes carries the launch number and tells what it replaced - nothing (GC_REQ, GC_SPEC_REQ: not this launch's
// business), GC_SPEC_MARK (GC_REQ_SPEC) or a pending GC_SPEC_REQ (GC_REQ_OVER).
constexpr int GC_SPEC_REQ = 8, GC_SPEC_MARK = 9, GC_REQ_SPEC = 10, GC_REQ_OVER = 12;
constexpr int GC_IDLE = 11;        // (a step, not a phase: a speculative marking that has nothing to do in this launch)
constexpr int GC_FOREIGN = 13;     // (a step, not a phase: a collection that launches with another number of collectors began - not touched)


// `gsv`: the wave's snapshot of the game's control block (word i in lane i), valid when nothing in this launch has changed
// the free-list words yet (the expansion's first attempt); has_gsv = false: read them from memory (sequential retries,
// update_root).
__device__ __forceinline__ void wave_new_nodes(const tm_store& S, const GP& P, WaveLds& L, int g, int n, int lane,
                                      int& r_idx, int& r_obs, bool has_gsv = false, int gsv = 0, bool barrier = false) {
    const bool act = lane < n;
    const uint32_t* my = L.slots[act ? lane : 0];
    const uint32_t mask = (uint32_t)S.table_cap - 1u;
    // with the snapshot the free-list tops are known now: fetch the n top entries of both lists under the table lookups
    int top_n = 0, top_o = 0;
    if (has_gsv) {
        const int nf = GSV(gsv, TM_GS_NFREE_NODE) - 1 - lane, of = GSV(gsv, TM_GS_NFREE_OBS) - 1 - lane;
        if (act && nf >= 0) top_n = P.fnode()[nf];
        if (act && of >= 0) top_o = P.fobs()[of];
    }
    uint64_t h = hash_game(my);
    // 1. candidates that are equal to an earlier candidate reuse its node (dict hit in the reference)
    int dup = lane;
    for (int b = 0; b < n; ++b) {
        uint64_t hb = rl_u64(h, b);
        if (act && dup == lane && b < lane && hb == h && eq_lds(my, L.slots[b], GAME_DW)) dup = b;
    }
    const bool uniq = act && dup == lane;
    // 2. transposition lookup by full state equality
    int found = 0;
    uint32_t ins = 0;
    bool full = false;
    if (uniq) found = table_find(P.ntab(), mask, h, my, P.game(), GAME_DW, ins, full);
    bool isnew = uniq && !found && !full;
    uint64_t need = __ballot(isnew);
    int cnt = __popcll(need);
    int nfree = has_gsv ? GSV(gsv, TM_GS_NFREE_NODE) : P.gs()[TM_GS_NFREE_NODE];
    if (__any(full)) { if (lane == 0) atomicOr(&P.gs()[TM_GS_ERR], TM_ERR_TABLE); }
    if (cnt > nfree) {
        // (a collection that completed earlier in this move - TM_GS_GC_IN_MOVE - has left nothing to reclaim: the search only
        // adds links below the root, so what was reachable then is reachable now and everything allocated since is linked.
        // The reference collects again and frees nothing, agents/agent.py:96-97; counted, not done.)
        const bool futile = has_gsv && GSV(gsv, TM_GS_GC_IN_MOVE) != 0 && GSV(gsv, TM_GS_GC_RETRY) == 0 && GSV(gsv, TM_GS_POOL_FULL) == 0;
        const bool no_more_gc = has_gsv && (GSV(gsv, TM_GS_GC_RETRY) != 0 || GSV(gsv, TM_GS_POOL_FULL) != 0 || futile);
        if (!no_more_gc) {
            // Pool exhausted: the reference reclaims unreachable nodes at exactly this pop
            // (agents/agent.py:96-97).  Candidates before the exhausting one are inserted first, exactly as
            // the sequential reference does, then a collection is requested and the rest is retried.
            // Handled by the caller-visible slow path below.
            r_idx = -1;
            r_obs = nfree;  // tells the caller how many can still be taken
            return;
        }
        // Exhausted although everything unreachable has just been reclaimed (or nothing can be before the root moves): the
        // reachable tree has outgrown the pool (the reference prints MAX_NODES EXCEEDED and runs into undefined behaviour,
        // agent.cpp:227-231).  What the sequential form does then, in one step: the first nfree new candidates in action
        // order get the remaining slots, the others the null node.  (Hundreds of games are in this state at any time once
        // the pools have filled: seven one-candidate calls per simulation made them the slowest waves of every launch.)
        isnew = isnew && __popcll(need & ((1ull << lane) - 1ull)) < nfree;
        need = __ballot(isnew);
        cnt = __popcll(need);
        if (lane == 0) {
            atomicOr(&P.gs()[TM_GS_ERR], TM_ERR_POOL); P.gs()[TM_GS_POOL_FULL] = 1;
            if (futile) P.gs()[TM_GS_N_GC] = GSV(gsv, TM_GS_N_GC) + 1;
        }
    }
    int idx = found;
    {
        const int rank = __popcll(need & ((1ull << lane) - 1ull));
        if (has_gsv) { const int v = (int)shfl_u32((uint32_t)top_n, rank); if (isnew) idx = v; }
        else if (isnew) idx = P.fnode()[nfree - 1 - rank];
    }
    if (cnt) {
        // lowest index ever allocated (GC skips clearing what was never written)
        int lo = 0x7FFFFFFF;
        for (int b = 0; b < n; ++b) lo = min(lo, ((need >> b) & 1ull) ? (int)rl_u32((uint32_t)idx, b) : 0x7FFFFFFF);
        const int low_seen = has_gsv ? GSV(gsv, TM_GS_LOW_NODE) : P.gs()[TM_GS_LOW_NODE];
        if (lane == 0) { P.gs()[TM_GS_NFREE_NODE] = nfree - cnt; if (lo < low_seen) P.gs()[TM_GS_LOW_NODE] = lo; }
    }
    table_insert_seq(P.ntab(), mask, need, n, lane, h, ins, idx, L.misc);
    // 3. observations of the new nodes (agents/agent.py:112-128)
    const bool ident = S.kind == TM_KIND_DIST;      // no projection: a node's statistics are its own (observation index = node index)
    uint32_t* ok = L.okeys[act ? lane : 0];
    uint64_t ho = 0;
    if (isnew && !ident) { pack_obs(my, ok); ho = hash_obs(ok); }
    wave_sync();
    int odup = lane;
    for (int b = 0; b < n; ++b) {
        uint64_t hb = rl_u64(ho, b);
        bool bnew = (need >> b) & 1ull;
        if (isnew && bnew && odup == lane && b < lane && hb == ho && eq_lds(ok, L.okeys[b], OBS_DW)) odup = b;
    }
    const bool ouniq = isnew && odup == lane && !ident;
    int ofound = 0;
    uint32_t oins = 0;
    bool ofull = false;
    if (ouniq) ofound = table_find(P.otab(), mask, ho, ok, P.okey(), OBS_DW, oins, ofull);
    bool onew = ouniq && !ofound && !ofull;
    uint64_t oneed = __ballot(onew);
    int ocnt = __popcll(oneed);
    int onfree = has_gsv ? GSV(gsv, TM_GS_NFREE_OBS) : P.gs()[TM_GS_NFREE_OBS];
    int o = ofound;
    {
        const int rank = __popcll(oneed & ((1ull << lane) - 1ull));
        if (has_gsv) { const int v = (int)shfl_u32((uint32_t)top_o, rank); if (onew) o = v; }
        else if (onew) o = P.fobs()[onfree - 1 - rank];
    }
    if (ocnt) {
        int lo = 0x7FFFFFFF;
        for (int b = 0; b < n; ++b) lo = min(lo, ((oneed >> b) & 1ull) ? (int)rl_u32((uint32_t)o, b) : 0x7FFFFFFF);
        const int olow_seen = has_gsv ? GSV(gsv, TM_GS_LOW_OBS) : P.gs()[TM_GS_LOW_OBS];
        if (lane == 0) { P.gs()[TM_GS_NFREE_OBS] = onfree - ocnt; if (lo < olow_seen) P.gs()[TM_GS_LOW_OBS] = lo; }
    }
    table_insert_seq(P.otab(), mask, oneed, n, lane, ho, oins, o, L.misc + 8);
    if (onew) {
        uint4* dst = reinterpret_cast<uint4*>(P.okey() + (size_t)o * OBS_DW);
        dst[0] = make_uint4(ok[0], ok[1], ok[2], ok[3]);
        dst[1] = make_uint4(ok[4], ok[5], ok[6], ok[7]);
        dst[2] = make_uint4(ok[8], ok[9], ok[10], ok[11]);
        // visit, value, variance = 0 and obs_arrays['end'] (agents/agent.py:123): slots are initialised when they are
        // handed out, so the GC does not have to clear the key / record streams of everything it frees
        *reinterpret_cast<uint4*>(P.stat() + (size_t)o * 4) = make_uint4((ok[11] & 1u) << 31, 0u, 0u, 0u);
        // ... and nothing is known about it yet (the evaluator's output per observation, TM_SIM_EVAL_NEEDED)
        if (S.obs_eval) *reinterpret_cast<uint4*>(S.obs_eval + ((size_t)g * P.n() + (size_t)o) * 4) = make_uint4(0u, 0u, 0u, 0u);
    }
    {   // same observation as an earlier new candidate
        int osrc = shfl_u32((uint32_t)o, odup);
        if (isnew && odup != lane) o = osrc;
    }
    if (ident && isnew) {
        // node_stats = (visit, mean, variance, M2) as floats, all zero; node_dist row zero (core_distributional.py's arrays)
        o = idx;
        *reinterpret_cast<uint4*>(P.stat() + (size_t)idx * 4) = make_uint4(0u, 0u, 0u, 0u);
        uint4* dr = reinterpret_cast<uint4*>(S.node_dist + ((size_t)g * P.n() + (size_t)idx) * TM_DIST_ROW);
#pragma unroll
        for (int t = 0; t < TM_DIST_ROW / 4; ++t) dr[t] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (isnew) {
        uint4* dst = reinterpret_cast<uint4*>(P.game() + (size_t)idx * GAME_DW);
        dst[0] = make_uint4(my[0], my[1], my[2], my[3]);
        dst[1] = make_uint4(my[4], my[5], my[6], my[7]);
        dst[2] = make_uint4(my[8], my[9], my[10], my[11]);
        dst[3] = make_uint4(my[12], my[13], my[14], my[15]);
        // the whole record: no children yet (slot triples and raw child row zero), then hdr = end flag of the node's
        // game, node_to_obs, arrays['score'][idx] = game.score (float32)
        static_assert(TM_REC_OBS == 29 && TM_REC_SCORE == 30 && TM_REC_HDR == 31 && TM_REC_DW == 32 && TM_KIDS_DW == 8,
                      "record layout");
        const uint4 z = make_uint4(0, 0, 0, 0);
        uint4* r = reinterpret_cast<uint4*>(P.rec() + (size_t)idx * TM_REC_DW);
        r[0] = z; r[1] = z; r[2] = z; r[3] = z; r[4] = z; r[5] = z; r[6] = z;
        r[7] = make_uint4(0u, (uint32_t)o, __float_as_uint((float)(int)my[14]), ((my[11] >> 8) & 1u) << 24);
        // the raw child row: no children, and the node's observation once more (the collectors' marker reads both in one row)
        uint4* kd = reinterpret_cast<uint4*>(P.kids() + (size_t)idx * TM_KIDS_DW);
        kd[0] = z; kd[1] = make_uint4(0u, 0u, 0u, (uint32_t)o);
    }
    if (uniq && found) o = (int)P.rec()[(size_t)found * TM_REC_DW + TM_REC_OBS];
    if (barrier) {
        // write barrier of a speculative marking (see GC_SPEC_MARK): every successor about to be linked is marked - a new
        // node, with its observation (the markers need not look at it: it has no children yet, and whatever is linked under
        // it later in this marking goes through this barrier too) - or handed to the markers
        const size_t bm_bytes = gc_bm_bytes(S.max_nodes);
        uint32_t* nmw = reinterpret_cast<uint32_t*>(S.gc_mark + (size_t)g * 2 * bm_bytes);
        uint32_t* omw = reinterpret_cast<uint32_t*>(S.gc_mark + (size_t)g * 2 * bm_bytes + bm_bytes);
        if (uniq && idx != 0) {
            const uint32_t w = (uint32_t)idx >> 5, bit = 1u << (idx & 31);
            if (isnew) {
                atomicOr(nmw + w, bit);
                atomicOr(omw + ((uint32_t)o >> 5), 1u << (o & 31));
            } else if (!(__hip_atomic_load(nmw + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) {
                // An existing node that is not marked (in the global bitmap: as of the last launch) is SENT to the owner of its
                // share of the index range like a child in another marking workgroup's share: its bit in the game's pending
                // bitmap, and - once that atomic has been performed (its value is back) - its chunk's flag.  An owner that sees
                // the flag finds the bit; an owner that has marked the node itself meanwhile drops it.  The node's MARK is the
                // owner's to set (a mark tells an owner "looked at, or pending with me"); what has been sent once is
                // remembered in a bitmap of the barrier's own (behind the pending bitmap).
                uint32_t* pdw = reinterpret_cast<uint32_t*>(S.gc_queue + (size_t)g * S.max_nodes);
                const uint32_t oldb = atomicOr(pdw + bm_bytes / 4 + w, bit), oldp = atomicOr(pdw + w, bit);
                asm volatile("" :: "v"(oldb), "v"(oldp) : "memory");
                if (!(oldb & bit)) {
                    const int ck = idx >> gc_chunk_log2(S.max_nodes);
                    atomicOr(&P.gs()[TM_GS_GC_FLAGS + (ck >> 5)], (int)(1u << (ck & 31)));
                }
            }
        }
    }
    {
        int isrc = shfl_u32((uint32_t)idx, dup), osrc = shfl_u32((uint32_t)o, dup);
        if (act && dup != lane) { idx = isrc; o = osrc; }
    }
    r_idx = idx;
    r_obs = o;
}

// ---------------------------------------------------------------------------------------------------
// expand (agents/agent.py:136-145): 7 successors of `leaf`, written into the leaf's record together
// with the unique-child list of get_unique_child_obs (core.h:111-144).
// Returns false when the node pool ran dry at one of the seven pops: the successors before it are inserted and linked
// (exactly the reference's state at the moment it calls remove_nodes, agents/agent.py:96-97), a collection is requested
// (TM_GS_GC_PHASE = 1) and the caller suspends the simulation; once the collection is complete the expansion is simply
// redone - the successors already inserted are transposition hits, the others pop from the rebuilt free list in order.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool wave_expand(const tm_store& S, const GP& P, WaveLds& L, int g, int lane, int leaf,
                                   uint32_t self_sc /* float bits of the leaf's own score */, uint32_t& hdr_out, int gsv,
                                   int gc_req_word /* (launch << 4) | GC_REQ: what a collection request looks like */) {
    if (lane < GAME_DW) L.slots[7][lane] = P.game()[(size_t)leaf * GAME_DW + lane];
    wave_sync();
    for (int t = lane; t < 7 * GAME_DW; t += 64) L.slots[t >> 4][t & 15] = L.slots[7][t & 15];
    wave_sync();
    EngCfg cfg{S.app, S.scoring, S.randomizer};
    const int drop = hard_drop_rows(L.slots[7], lane);       // for the successor of action 3: 21 lanes instead of a 20-step loop in one
    // the piece a lock spawns: the same for every successor (seed and piece count are the parent's), the lanes' work (engine.h)
    const int next_piece = wave_piece_at(L.slots[7][13], L.slots[7][12], S.randomizer, lane);
    if (lane < 7) {
        Piece p;
        load_fields(L.slots[lane], p);
        play(reinterpret_cast<uint16_t*>(L.slots[lane]), p, cfg, lane, nullptr, drop, next_piece);
        store_fields(L.slots[lane], p);
    }
    wave_sync();
    int idx, o;
    const bool barrier = (gc_req_word & 15) == GC_REQ_SPEC;      // the launch began in GC_SPEC_MARK
    wave_new_nodes(S, P, L, g, 7, lane, idx, o, true, gsv, barrier);
    if (idx < 0) {
        // slow path: sequential new_node with a GC at the exhausting pop (rare: once per ~50 moves)
        int out_idx = 0, out_o = 0;
        for (int a = 0; a < 7; ++a) {
            // move candidate a into slot 0 of a scratch view: reuse slot 7 (the parent copy is no longer needed)
            if (lane < GAME_DW) L.slots[7][lane] = L.slots[a][lane];
            wave_sync();
            // run a 1-candidate new_node on slot 7 by temporarily swapping it with slot 0
            uint32_t keep = 0;
            if (lane < GAME_DW) { keep = L.slots[0][lane]; L.slots[0][lane] = L.slots[7][lane]; }
            wave_sync();
            int i1, o1;
            wave_new_nodes(S, P, L, g, 1, lane, i1, o1, false, 0, barrier);
            if (i1 < 0) {
                if (!P.gs()[TM_GS_GC_RETRY] && !P.gs()[TM_GS_POOL_FULL]) {
                    if (lane < GAME_DW) L.slots[0][lane] = keep;
                    if (lane == 0) {
                        P.gs()[TM_GS_GC_REQ_AT] = gc_req_word >> 4;
                        atomicExch(&P.gs()[TM_GS_GC_PHASE], gc_req_word);
                        gc_active_set(P);
                        P.gs()[TM_GS_GC_RETRY] = 1;
                    }
                    wave_sync();
                    return false;
                }
                // The pool is exhausted although everything unreachable has just been reclaimed: the reachable tree has
                // outgrown it (the reference prints MAX_NODES EXCEEDED and runs into undefined behaviour, agent.cpp:227-231).
                // Nothing becomes unreachable before the root moves, so no further collection is attempted until then
                // (TM_GS_POOL_FULL; without it every remaining simulation of the move would sweep the whole pool in vain).
                if (lane == 0) { atomicOr(&P.gs()[TM_GS_ERR], TM_ERR_POOL); P.gs()[TM_GS_POOL_FULL] = 1; }
                i1 = 0; o1 = 0;
            }
            i1 = (int)rl_u32((uint32_t)i1, 0);
            o1 = (int)rl_u32((uint32_t)o1, 0);
            if (lane < GAME_DW) L.slots[0][lane] = keep;
            wave_sync();
            if (lane == a) { out_idx = i1; out_o = o1; }
            // the reference links child[idx][a] right after each new_node (agent.py:145), which is what
            // makes earlier successors reachable for a later GC
            if (lane == 0) P.kids()[(size_t)leaf * TM_KIDS_DW + a] = (uint32_t)i1;
            wave_sync();
        }
        idx = out_idx;
        o = out_o;
    }
    uint32_t sbits = __float_as_uint((float)(int)L.slots[lane < 7 ? lane : 0][14]);
    if (lane < 7) P.kids()[(size_t)leaf * TM_KIDS_DW + lane] = (uint32_t)idx;    // raw children, action order
    // get_unique_child_obs (core.h:126-142), evaluated once here because its inputs never change.  Lane a owns
    // action a: every lane scans the actions in order, finds the first action with its observation ("first") and,
    // from there on, keeps the strictly-greater score (the reference's replacement rule).  The first lane of each
    // group is its leader; leaders are compacted in first-seen order = the reference's c_nodes / c_obs order, and
    // write their (representative child, observation, its score) triple into the record's slot.
    {
        const bool act7 = lane < 7;
        const uint32_t my_c = act7 ? (uint32_t)idx : 0u, my_o = act7 ? (uint32_t)o : 0u;
        const float my_s = __uint_as_float(sbits);
        int first = lane;
        uint32_t best_c = my_c;
        float best_s = my_s;
        bool seen_first = false;
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            const uint32_t cb = rl_u32(my_c, b), ob = rl_u32(my_o, b);
            const float sb = rl_f32(my_s, b);
            const bool same = act7 && cb != 0 && my_c != 0 && ob == my_o;
            if (same && !seen_first) { first = b; seen_first = true; best_c = cb; best_s = sb; }
            else if (same && sb > best_s) { best_c = cb; best_s = sb; }
        }
        const bool leader = act7 && my_c != 0 && first == lane;
        const uint64_t lead_mask = __ballot(leader);
        const int nu = __popcll(lead_mask);
        const int slot = __popcll(lead_mask & ((1ull << lane) - 1ull));
        const uint32_t hdr = (uint32_t)nu | (1u << 25);
        uint32_t* r = P.rec() + (size_t)leaf * TM_REC_DW;
        if (lane < 7) { L.misc[16 + lane] = 0; L.misc[24 + lane] = 0; L.misc[32 + lane] = 0; }
        wave_sync();
        if (leader) {
            // (child, observation, child score, own score): the walk's lane group for this slot needs nothing else
            reinterpret_cast<uint4*>(r)[slot] = make_uint4(best_c, my_o, __float_as_uint(best_s), self_sc);
            L.misc[16 + slot] = best_c;                 // slot-ordered copies for the evaluation requests
            L.misc[24 + slot] = my_o;
            L.misc[32 + slot] = __float_as_uint(best_s);
        }
        if (lane == 0) { L.misc[56] = hdr; r[TM_REC_HDR] = hdr; }
    }
    wave_sync();
    hdr_out = L.misc[56];
    return true;
}

// ---------------------------------------------------------------------------------------------------
// Welford update of one observation (core.h:248-257); fp64 arithmetic, fp32 storage, no contraction
// ---------------------------------------------------------------------------------------------------
// statistics record of an observation: visit | end << 31, value, variance and the exploration term of policy_clt
// (core.h:98: sqrtf(variance / (float)visit), the same two correctly rounded float operations, evaluated here once
// per update instead of once per child per tree level in the walk)
__device__ __forceinline__ void store_stat(uint32_t* st, uint32_t old_x, int visit, float value, float variance) {
    const float ratio = variance / (float)visit;
    const float root = sqrtf(ratio);
    *reinterpret_cast<uint4*>(st) = make_uint4((uint32_t)visit | (old_x & 0x80000000u), __float_as_uint(value),
                                               __float_as_uint(variance), __float_as_uint(root));
}
__device__ __forceinline__ void welford_f64(uint32_t* st, uint4 s /* the record at st, loaded by the caller */, double x, double var_in) {
    int visit = (int)(s.x & 0x7FFFFFFFu);
    float value = __uint_as_float(s.y), variance = __uint_as_float(s.z);
    if (visit == 0) {
        value = (float)x;
        variance = (float)var_in;
    } else {
        double delta = x - (double)value;
        value = (float)((double)value + delta / (double)(visit + 1));
        double delta2 = x - (double)value;
        double prod = delta * delta2;
        variance = (float)((double)variance + (prod - (double)variance) / (double)(visit + 1));
    }
    store_stat(st, s.x, visit + 1, value, variance);
}
// agent.cpp:496-513: the carried value is a float
__device__ __forceinline__ void welford_f32carry(uint32_t* st, uint4 s /* the record at st, loaded by the caller */, float x, float var_in) {
    int visit = (int)(s.x & 0x7FFFFFFFu);
    float value = __uint_as_float(s.y), variance = __uint_as_float(s.z);
    if (visit == 0) {
        value = x;
        variance = var_in;
    } else {
        double delta = (double)(x - value);
        value = (float)((double)value + delta / (double)(visit + 1));
        double delta2 = (double)(x - value);
        double prod = delta * delta2;
        variance = (float)((double)variance + (prod - (double)variance) / (double)(visit + 1));
    }
    store_stat(st, s.x, visit + 1, value, variance);
}

// backup of the stored trace.  Entries are handled 64 at a time from the leaf end; the carried value
// is a pure function of the scores, so every lane replays the recurrence and keeps its own entry's
// value; the observation updates are independent unless an observation repeats in the trace, which
// ENGINE_SPEC.md rules out for app == 1 (checked, with a sequential fallback, when app > 1).
template <bool FLOAT_CARRY>
__device__ __forceinline__ void wave_backup_trace_t(const tm_store& S, const GP& P, int lane, int len, double v0, double var0) {
    const double gamma = S.gamma;
    double V = v0;
    float Vf = (float)v0;
    const float varf = (float)var0;
    for (int base = 0; base < len; base += 64) {
        int cnt = min(64, len - base);
        int i = len - 1 - base - lane;
        uint4 e = make_uint4(0, 0, 0, 0);      // (-, observation, score bits, -)
        if (lane < cnt) e = reinterpret_cast<const uint4*>(P.trace())[i];
        // the observation's statistics are requested before the recurrence below and used after it: one load (as separate
        // words the optimizer fetched the visit count, branched on it and fetched the rest - two trips), under the loop
        uint32_t* st = P.stat() + (size_t)e.y * 4;
        uint4 sv = *reinterpret_cast<const uint4*>(st);      // lanes >= cnt: entry 0 (the null observation), never stored
        double x = 0;
        float xf = 0;
        // the carried value is a pure function of the scores: every lane replays the recurrence and keeps the value of its
        // own entry.  (The carry type is a template parameter: as a run-time flag it put four branches into this loop.)
        for (int j = 0; j < cnt; ++j) {
            const float sj = rl_f32(__uint_as_float(e.z), j);   // wave-uniform source lane
            if (FLOAT_CARRY) {
                Vf = Vf - sj;
                xf = (lane == j) ? Vf : xf;
                const double t = gamma * (double)Vf;
                Vf = (float)(t + (double)sj);
            } else {
                V = V - (double)sj;
                x = (lane == j) ? V : x;
                const double t = gamma * V;
                V = t + (double)sj;
            }
        }
        asm volatile("" : "+v"(sv.x), "+v"(sv.y), "+v"(sv.z));
        if (lane < cnt) {
            if (FLOAT_CARRY) welford_f32carry(st, sv, xf, varf);
            else welford_f64(st, sv, x, var0);
        }
    }
}
__device__ inline void wave_backup_trace(const tm_store& S, const GP& P, int lane, int len, double v0, double var0,
                                         bool float_carry) {
    if (float_carry) wave_backup_trace_t<true>(S, P, lane, len, v0, var0);
    else wave_backup_trace_t<false>(S, P, lane, len, v0, var0);
}
// strictly sequential form (one lane), used when an observation may repeat inside the trace
__device__ inline void lane_backup_trace_seq(const tm_store& S, const GP& P, int len, double v0, double var0,
                                             bool float_carry) {
    const double gamma = S.gamma;
    double V = v0;
    float Vf = (float)v0;
    for (int i = len - 1; i >= 0; --i) {
        uint4 e = reinterpret_cast<const uint4*>(P.trace())[i];
        float sj = __uint_as_float(e.z);
        uint32_t* st = P.stat() + (size_t)e.y * 4;
        const uint4 sv = *reinterpret_cast<const uint4*>(st);
        if (float_carry) {
            Vf = Vf - sj;
            welford_f32carry(st, sv, Vf, (float)var0);
            double t = gamma * (double)Vf;
            Vf = (float)(t + (double)sj);
        } else {
            V = V - (double)sj;
            welford_f64(st, sv, V, var0);
            double t = gamma * V;
            V = t + (double)sj;
        }
        __threadfence_block();
    }
}

__device__ inline bool trace_has_repeat(const GP& P, int lane, int len, const int* extra, int n_extra) {
    // O(len^2/64) comparison of observation indices; only called when app > 1
    bool rep = false;
    for (int base = 0; base < len; base += 64) {
        int i = base + lane;
        uint32_t oi = (i < len) ? P.trace()[(size_t)i * 4 + 1] : 0xFFFFFFFFu;
        for (int j = 0; j < len; ++j) {
            uint32_t oj = P.trace()[(size_t)j * 4 + 1];
            if (i < len && j != i && oj == oi) rep = true;
        }
        for (int j = 0; j < n_extra; ++j)
            if (i < len && (uint32_t)extra[j] == oi) rep = true;
    }
    return __any(rep);
}

// The evaluator's output for the leaf's observation, filed for later leaves with the same observation (TM_SIM_EVAL_NEEDED: the
// built-in value net is a function of the rendered observation and of the weights, whose version is S.eval_epoch).  Only a
// request that was really posted is filed (K_EVAL == 1; a leaf answered from the cache has K_EVAL == 0).
__device__ __forceinline__ void eval_cache_store(const tm_store& S, const GP& P, int lane, int gsv, int eflags, int leaf_end) {
    if (!(eflags & TM_SIM_EVAL_NEEDED) || !S.obs_eval || leaf_end || GSV(gsv, TM_GS_K_EVAL) != 1) return;
    if (lane == 0)
        *reinterpret_cast<uint4*>(S.obs_eval + ((size_t)P.g * P.n() + (size_t)GSV(gsv, TM_GS_LEAF_OBS)) * 4) =
            make_uint4(__float_as_uint(P.eval_v()[0]), __float_as_uint(P.eval_var()[0]), (uint32_t)S.eval_epoch, 0u);
}

// ---------------------------------------------------------------------------------------------------
// the back half of a simulation: ValueSim.py:83-94 / ValueSimLP.py:59-70 / agent.cpp:432-446,458
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sim_back(const tm_store& S, const GP& P, WaveLds& L, int lane, int gsv, int eflags) {
    const int len = GSV(gsv, TM_GS_TRACE_LEN);
    const int leaf_end = GSV(gsv, TM_GS_LEAF_END);
    const int k = GSV(gsv, TM_GS_K_EVAL);
    const int leaf_score = GSV(gsv, TM_GS_LEAF_SCORE);
    const int kind = S.kind;
    const bool fcarry = (kind == TM_KIND_CPPAGENT_LP || kind == TM_KIND_CPPAGENT || kind == TM_KIND_VANILLA_C);
    double v0 = 0, var0 = 0;
    bool seq = false;
    if (kind == TM_KIND_VANILLA) {
        v0 = (double)leaf_score;   // the rollout's final score (or the terminal leaf's own), Vanilla.py:54,59
        var0 = leaf_end ? 0.0 : 1e3;
    } else if (kind == TM_KIND_VANILLA_C) {
        v0 = (double)(float)leaf_score;   // agent.cpp:452-454: float _val = (game.score, 1e5)[0]; terminal leaf: (float)games[leaf].score
        var0 = leaf_end ? 0.0 : 1e5;
    } else if (kind == TM_KIND_VALUESIM) {
        v0 = (double)leaf_score;   // Python int score + np.float32 v under numpy 1.17 = float64
        if (!leaf_end) { v0 = v0 + (double)P.eval_v()[0]; var0 = (double)P.eval_var()[0]; }
        eval_cache_store(S, P, lane, gsv, eflags, leaf_end);
    } else if (kind == TM_KIND_CPPAGENT) {
        float ls = (float)leaf_score;   // score[trace.back()] is a float array element
        if (!leaf_end) { v0 = (double)(ls + P.eval_v()[0]); var0 = (double)P.eval_var()[0]; }
        else v0 = (double)ls;
        eval_cache_store(S, P, lane, gsv, eflags, leaf_end);
    } else {
        // leaf-parallel: first-visit initialisation of the unique children, then the averaged target
        if (S.app > 1) seq = trace_has_repeat(P, lane, len, P.leaf() + 7, k);
        if (k > 0) {
            int co = (lane < k) ? P.leaf()[7 + lane] : 0;
            float cs = (lane < k) ? __int_as_float(P.leaf()[14 + lane]) : 0.f;
            uint4 st = make_uint4(0, 0, 0, 0);
            if (lane < k) st = *reinterpret_cast<const uint4*>(P.stat() + (size_t)co * 4);
            float cval = __uint_as_float(st.y), cvar = __uint_as_float(st.z);
            if (lane < k && (st.x & 0x7FFFFFFFu) == 0) {
                // core.h:344-352 tests end[child node] with the never-written arrays['end'] (ValueSimLP.py:25):
                // always false.  agent.cpp:538 tests end_obs[o].
                bool e = (kind == TM_KIND_CPPAGENT_LP) ? ((st.x >> 31) != 0) : false;
                cval = e ? 0.f : P.eval_v()[lane];
                cvar = e ? 0.f : P.eval_var()[lane];
                store_stat(P.stat() + (size_t)co * 4, st.x, 1, cval, cvar);
            }
            double vt = 0, vart = 0;
            for (int i = 0; i < k; ++i) {
                float vi = rl_f32(cval, i);
                float ri = rl_f32(cvar, i);
                float si = rl_f32(cs, i);
                double gv = S.gamma * (double)vi;
                vt = vt + ((double)si + gv);
                vart = vart + (double)ri;
            }
            vt = vt / (double)k;
            if (kind == TM_KIND_VALUESIM_LP) vart = vart * (S.gamma * S.gamma / (double)k);
            else vart = vart / (double)k;
            if (fcarry) { v0 = (double)(float)vt; var0 = (double)(float)vart; }
            else { v0 = vt; var0 = vart; }
        } else {
            v0 = (double)(float)leaf_score;   // score_uc(trace[-1]) / (float)games[leaf].score
            var0 = 0;
        }
        __threadfence_block();
    }
    if (S.app > 1 && kind != TM_KIND_VALUESIM_LP && kind != TM_KIND_CPPAGENT_LP)
        seq = trace_has_repeat(P, lane, len, nullptr, 0);
    if (seq) { if (lane == 0) lane_backup_trace_seq(S, P, len, v0, var0, fcarry); }
    else wave_backup_trace(S, P, lane, len, v0, var0, fcarry);
    if (lane == 0) { P.gs()[TM_GS_PENDING] = 0; P.gs()[TM_GS_N_SIMS] = GSV(gsv, TM_GS_N_SIMS) + 1; }
}

// ---------------------------------------------------------------------------------------------------
// second stage of the front half: expansion of the leaf and the evaluation requests.  Runs right after the
// walk, or again from the control block when the expansion had to wait for a collection.
// ---------------------------------------------------------------------------------------------------
template <bool VANILLA>
__device__ __forceinline__ void wave_front_finish(const tm_store& S, const GP& P, WaveLds& L, int g, int lane, int leaf,
                                                  int leaf_end, uint32_t self_o, uint32_t self_sc, int gsv, int gc_req_word,
                                                  int eflags) {
    const long long tc0 = __builtin_readcyclecounter();
    const int kind = S.kind;
    int k_eval = 0;                 // what the backup finds in TM_GS_K_EVAL: the leaf's unique children (leaf-parallel kinds), else requests posted
    int n_post = 0, n_skip = 0, n_cached = 0;
    bool post = false;              // this lane's request slot carries a request
    uint32_t post_obs = 0;
    // single-leaf kinds under TM_SIM_EVAL_NEEDED: what the evaluator returned for this observation when it was a leaf before
    // (another node, same board and piece), requested here so that it arrives under the expansion
    const bool use_cache = !VANILLA && !leaf_end && (eflags & TM_SIM_EVAL_NEEDED) && S.obs_eval && S.eval_epoch > 0 &&
                           (kind == TM_KIND_VALUESIM || kind == TM_KIND_CPPAGENT);
    uint4 ce = make_uint4(0u, 0u, 0u, 0u);
    if (use_cache) ce = *reinterpret_cast<const uint4*>(S.obs_eval + ((size_t)g * P.n() + (size_t)self_o) * 4);
    if (!leaf_end) {
        uint32_t lh;
        if (!wave_expand(S, P, L, g, lane, leaf, self_sc, lh, gsv, gc_req_word)) {
            if (lane < S.eval_slots) P.eval_obs()[lane] = 0;
            if (lane == 0) P.gs()[TM_GS_PENDING] = 2;
            return;
        }
        if (VANILLA) {
            k_eval = 0;
            if (lane < S.eval_slots) P.eval_obs()[lane] = 0;
        } else if (kind == TM_KIND_VALUESIM || kind == TM_KIND_CPPAGENT || kind == TM_KIND_DIST) {
            // (TM_KIND_DIST: the request names the leaf NODE - identity projection)
            // filed under the same weights: the evaluator would return the same two floats
            const bool hit = use_cache && (int)ce.z == S.eval_epoch;
            if (hit && lane == 0) { P.eval_v()[0] = __uint_as_float(ce.x); P.eval_var()[0] = __uint_as_float(ce.y); }
            n_cached = hit ? 1 : 0;
            k_eval = n_post = hit ? 0 : 1;
            post = lane == 0 && !hit;
            post_obs = self_o;
            if (lane == 0) P.eval_obs()[0] = hit ? 0 : (int)self_o;
        } else {
            // unique children of the freshly expanded leaf (ValueSimLP.py:55 / agent.cpp:424).  The backup uses a child's
            // output only to initialise an observation on its first visit (core.h:341-350; agent.cpp:536-545: unless it is a
            // finished one): those are counted, and under TM_SIM_EVAL_NEEDED only those are posted.
            const int nu = (int)(lh & 7u);
            k_eval = nu;
            uint32_t sx = 0;
            if (lane < nu) sx = P.stat()[(size_t)L.misc[24 + lane] * 4];
            const bool used = lane < nu && (sx & 0x7FFFFFFFu) == 0u && !(kind == TM_KIND_CPPAGENT_LP && (sx >> 31) != 0u);
            post = lane < nu && (used || !(eflags & TM_SIM_EVAL_NEEDED));
            n_skip = nu - __popcll(__ballot(used));
            n_post = __popcll(__ballot(post));
            post_obs = lane < nu ? L.misc[24 + lane] : 0u;
            if (lane < 7) {
                bool on = lane < nu;
                P.leaf()[lane] = on ? (int)L.misc[16 + lane] : 0;
                P.leaf()[7 + lane] = on ? (int)L.misc[24 + lane] : 0;
                P.leaf()[14 + lane] = on ? (int)L.misc[32 + lane] : 0;
                P.eval_obs()[lane] = post ? (int)L.misc[24 + lane] : 0;
            }
        }
        if (n_post > 0) {
            // the dense request list (include/tetris_mcts_hip.h, eval_list): one atomic per game and launch
            const int slots = S.eval_slots, segs = TM_EVAL_SEGS(S.n_games), seg = g % segs;
            int base = 0;
            if (lane == 0) base = atomicAdd(&S.eval_cnt[(size_t)seg * 2 + S.eval_parity], n_post);
            base = (int)rl_u32((uint32_t)base, 0);
            const uint64_t pm = __ballot(post);
            if (post) {
                const int d = base + __popcll(pm & ((1ull << lane) - 1ull));
                const int at = (seg + segs * (d / slots)) * slots + d % slots;
                // (a segment overflows only if the caller did not flip eval_parity between two launches: the counters then
                // keep growing.  Never write past the list - the game is flagged instead.)
                if (at < S.n_games * slots) reinterpret_cast<int2*>(S.eval_list)[at] = make_int2(g * slots + lane, (int)post_obs);
                else atomicOr(&P.gs()[TM_GS_ERR], TM_ERR_EVAL_LIST);
            }
        }
        if (lane == 0) P.gs()[TM_GS_N_EXPAND] = GSV(gsv, TM_GS_N_EXPAND) + 1;
    } else {
        if (lane < S.eval_slots) P.eval_obs()[lane] = 0;
    }
    if (lane == 0) {
        int32_t* gs = P.gs();
        gs[TM_GS_CYC_EXPAND] = (int)((long long)__builtin_readcyclecounter() - tc0);
        gs[TM_GS_PENDING] = 1;
        gs[TM_GS_GC_RETRY] = 0;
        gs[TM_GS_K_EVAL] = k_eval;
        gs[TM_GS_N_EVAL] = GSV(gsv, TM_GS_N_EVAL) + n_post;
        gs[TM_GS_LEAF_OBS] = (int)self_o;
        if (n_skip) gs[TM_GS_N_EVAL_SKIP] = GSV(gsv, TM_GS_N_EVAL_SKIP) + n_skip;
        if (n_cached) gs[TM_GS_N_EVAL_CACHED] = GSV(gsv, TM_GS_N_EVAL_CACHED) + n_cached;
        // nearly out of nodes: have the tree marked while the game goes on (GC_SPEC_*), unless a collection could not help
        // ... and only if the pool can run dry in THIS move (seven nodes a simulation at most): update_root drops the marks
        if (S.gc_spec_nodes > 0 && GSV(gsv, TM_GS_GC_PHASE) == 0 && !leaf_end && !GSV(gsv, TM_GS_POOL_FULL) && !gs[TM_GS_POOL_FULL]) {
            const int nf = gs[TM_GS_NFREE_NODE], sims_left = GSV(gsv, TM_GS_SIM_TARGET) - GSV(gsv, TM_GS_SIM_STARTED);
            if (nf < S.gc_spec_nodes && nf < 7 * sims_left) { gs[TM_GS_GC_PHASE] = (gc_req_word & ~15) | GC_SPEC_REQ; gc_active_set(P); }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// the front half: select_trace_obs (core.h:167-224), then expansion and evaluation requests
// ---------------------------------------------------------------------------------------------------
template <bool VANILLA>
__device__ __forceinline__ void wave_sim_front(const tm_store& S, const GP& P, WaveLds& L, MtLds* M, int g, int lane, int gsv,
                                               int gc_req_word, int eflags) {
    const long long tc_start = __builtin_readcyclecounter();
    if (lane < 32) L.misc[lane] = S.rng[(size_t)g * 32 + lane];      // glibc rand() state (31 words), used by check_low only
    wave_sync();
    int rng_pos = GSV(gsv, TM_GS_RNG_POS);
    const int rng_pos0 = rng_pos;
    int idx = GSV(gsv, TM_GS_ROOT);
    int len = 0;
    uint32_t hdr = 0, self_o = 0;
    const int low = S.low;
    int nq_fallback = 0;
    bool overflow = false;
    // Node record = eight 16-byte pieces: pieces 0..6 are the unique children in selection order (child, observation,
    // child score, own score), piece 7 is (predicted child, own observation, own score, header).  The wave is eight
    // 8-lane groups: every lane of group j holds piece j of the current node, so group j owns unique child j.
    //
    // The walk is a dependent chain (the child to descend into is known only after the children's statistics are
    // here), so it is software-pipelined on a PREDICTION: every node remembers the slot its last walk took (header
    // bits 4-6) and that child's index (piece 7, word 0).  One round of loads per level fetches the statistics of the
    // node's children AND the predicted child's record - piece j straight into group j, where it will be needed - and as
    // soon as that record is here the NEXT level's round (the predicted child's children's statistics, its own predicted
    // child) is issued, BEFORE this level's selection arithmetic.  The arithmetic (check_low / policy_clt, unchanged, on
    // the true statistics) then runs under the next level's memory latency.  If it selects another child than the
    // predicted one (a few per cent of the levels; always at a freshly expanded node), the speculative loads are dropped,
    // the node's prediction is rewritten and the walk pays one extra round trip.  Results do not depend on predictions.
    // Nothing else is stored to global memory inside the walk (stores share the load counter on gfx9); the trace goes
    // through LDS and is flushed 64 entries at a time with coalesced stores.
    const int grp = lane >> 3;
    const uint32_t grp16 = (uint32_t)grp * 16u;
    const __amdgpu_buffer_rsrc_t rec_rs = make_rsrc(P.rec(), P.n() * (TM_REC_DW * 4u));
    const __amdgpu_buffer_rsrc_t stat_rs = make_rsrc(P.stat(), P.n() * 16u);
    // per-lane constants of the loop: the visit mask drops the end flag and makes group 7 (the node's own piece, whose
    // statistics load is the node's own observation) count as "no child"; a NaN score wins at slot 0 and loses elsewhere
    const uint32_t vmask = lane < 56 ? 0x7FFFFFFFu : 0u;
    const int nan_key = grp == 0 ? 0x7FFFFFFF : (int)0x80000000;
    const uint64_t lanes_lt56 = 0x00FFFFFFFFFFFFFFull;
    // trace: lane 56 (group 7 holds the node's own observation and score) appends 8 bytes per level; the other lanes
    // store theirs to a dummy slot, so the walk needs no exec masking for it
    uint4* tp = (lane == 56) ? &L.tbuf[0] : &L.tdummy[lane];
    const int tinc = (lane == 56) ? 1 : 0;
    int flushed = 0;          // entries [0, flushed) of the trace are in global memory (set to the verified prefix below)
    auto flush_trace = [&](int upto) {   // entries [flushed, upto) from LDS to global, coalesced
        for (int base = flushed; base < upto; base += 64) {
            int i = base + lane;
            if (i < upto) reinterpret_cast<uint4*>(P.trace())[i] = L.tbuf[i - flushed];
        }
        flushed = upto;
    };
    // (float)norm_quantile(n): wave-uniform index into a table nobody writes = one s_load_dword (scalar cache) at a
    // 32-bit offset; the inline form keeps the address arithmetic to one shift
    const uint64_t nq_base = (uint64_t)(uintptr_t)S.nq_table;
    const int nq_size = S.nq_size, max_trace = S.max_trace;
    int n_miss = 0;
#define TM_NQ_LOOKUP(CB, N) asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(CB) : "s"(nq_base), "s"((N) << 2));
    // ---- the part of the previous walk that is still the selected path, verified in parallel ----
    // Nine levels out of ten of a walk repeat the game's previous walk (profiles/r03_walk_prefix_before.jsonl: 92 % of all
    // levels; five levels remain after the first difference, for the longest walks as for the mean).  The previous walk is
    // in the trace buffer (word 0 of an entry = the node), its nodes' records are immutable once expanded, and whether
    // level i still selects path[i + 1] depends only on the children's CURRENT statistics - not on the levels above it.
    // So the levels are checked independently, eight at a time (one 8-lane group per level, lane t = unique child t),
    // four such chunks per round of loads: records, then statistics, then the norm_quantile words, all in flight together,
    // instead of one dependent round trip per level.  The arithmetic is policy_clt's, operation for operation; a level
    // with an under-visited child (check_low would draw from rand()), a table miss or another selection ends the prefix,
    // and the pipelined serial walk below takes over at that level (its entries [0, k0) of the trace are already in place).
    int k0 = 0;
    {
        const int prev_len = GSV(gsv, TM_GS_TRACE_LEN);
        const uint32_t* tr = P.trace();
        const int nver = prev_len - 1;          // levels 0 .. nver - 1 have a recorded successor
        // (word 0 of entry 0 = the root of the previous walk: after update_root the path does not start at this root)
        uint32_t pa = nver > 0 && lane < prev_len ? tr[(size_t)lane * 4] : 0u;
        if (nver > 0 && rl_u32(pa, 0) == (uint32_t)idx) {
            const int t8 = lane & 7;
            const uint32_t t16 = (uint32_t)t8 * 16u;
            const uint32_t vm8 = t8 < 7 ? 0x7FFFFFFFu : 0u;
            const int nan8 = t8 == 0 ? 0x7FFFFFFF : (int)0x80000000;
            const float* nqt = S.nq_table;
            int base = 0;
            bool stop = false;
            uint32_t pb = 0;
            while (!stop) {
                // path[base + lane] and path[base + lane + 1]: 64 levels per coalesced load
                const int li = base + lane;
                if (base != 0) pa = li < prev_len ? tr[(size_t)li * 4] : 0u;
                pb = li + 1 < prev_len ? tr[(size_t)(li + 1) * 4] : 0u;
                k0 = min(nver, base + 64);
                for (int c0 = 0; c0 < 64 && base + c0 < nver; c0 += 32) {
                    uint4 rc[4], sc[4];
                    uint32_t ex[4], nsum[4];
                    float cb[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int src = (c0 + 8 * c + grp) * 4;           // the lane that holds this group's level
                        const uint32_t node = bperm_u32(src, pa);
                        ex[c] = bperm_u32(src, pb);
                        rc[c] = buf_ld16(rec_rs, node * (TM_REC_DW * 4u) + t16);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) sc[c] = buf_ld16(stat_rs, t8 < 7 ? rc[c].y * 16u : 0u);   // lane 7 of a group holds the node's own piece
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        uint32_t n = sc[c].x & vm8;                     // empty slots read the null observation: 0 visits
                        n += dpp_x1(n); n += dpp_x2(n); n += dpp_hm(n);
                        nsum[c] = n;
                        cb[c] = nqt[min(n, (uint32_t)(nq_size - 1))];
                    }
                    uint64_t bad[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bool act = base + c0 + 8 * c + grp < nver;
                        const bool exists = rc[c].x != 0u && t8 < 7;
                        const int visit = (int)(sc[c].x & vm8);
                        const uint64_t lowb = __builtin_amdgcn_ballot_w64(exists && visit < low);
                        const bool low_any = ((lowb >> (grp * 8)) & 0xFFull) != 0ull;
                        const float t1 = __uint_as_float(sc[c].y) + __uint_as_float(rc[c].z);
                        const float val = t1 - __uint_as_float(rc[c].w);
                        const float prod = cb[c] * __uint_as_float(sc[c].w);
                        const float q = (val + prod) + 0.0f;
                        const uint32_t qb = __float_as_uint(q);
                        int key = (int)(qb ^ (((uint32_t)((int)qb >> 31)) >> 1));
                        key = (q != q) ? nan8 : key;
                        key = exists ? key : (int)0x80000000;
                        int km = key;
                        km = max(km, (int)dpp_x1((uint32_t)km));
                        km = max(km, (int)dpp_x2((uint32_t)km));
                        km = max(km, (int)dpp_hm((uint32_t)km));
                        const uint64_t eq = __builtin_amdgcn_ballot_w64(key == km);
                        const int slot = __builtin_ctz((uint32_t)((eq >> (grp * 8)) & 0xFFull) | 0x100u);
                        const uint32_t csel = bperm_u32(((lane & ~7) + (slot & 7)) * 4, rc[c].x);
                        const bool ok = !act || (csel == ex[c] && !low_any && nsum[c] < (uint32_t)nq_size);
                        bad[c] = ~__builtin_amdgcn_ballot_w64(ok) & 0x0101010101010101ull;
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (!stop && bad[c] != 0ull) { k0 = base + c0 + 8 * c + (__builtin_ctzll(bad[c]) >> 3); stop = true; }
                    if (stop) break;
                }
                if (stop || base + 64 >= nver) break;
                base += 64;
            }
            // the node of level k0: path[k0]
            idx = (int)(k0 == base ? rl_u32(pa, 0) : rl_u32(pb, k0 - base - 1));
            len = k0;
        }
    }
    const long long tc_ver = __builtin_readcyclecounter();
    // three record register sets and two statistics sets rotate through the roles (current node, predicted child,
    // predicted grandchild) / (this level, next level): the loop body is instantiated six times instead of moving
    // eleven registers per level.  p0..p2 = predicted child (piece 7, word 0) of the node held in r0..r2.
    uint4 r0 = buf_ld16(rec_rs, (uint32_t)idx * (TM_REC_DW * 4u) + grp16), r1, r2;
    uint32_t p0 = rl_u32(r0.x, 56), p1 = 0, p2 = 0;
    uint4 s0 = buf_ld16(stat_rs, r0.y * 16u), s1 = s0;
    r1 = buf_ld16(rec_rs, p0 * (TM_REC_DW * 4u) + grp16);
    r2 = r1;
    int cur_node = idx;
    flushed = len;
    int room = min(TRACE_LDS, max_trace - len);
#define TM_WALK_LEVEL(RC, PC, SC, RN, PN, SN, RNN)                                                                      \
    {                                                                                                                   \
        if (__builtin_expect(room == 0, 0)) {     /* the LDS trace buffer is full, or the trace is */                   \
            if (len >= max_trace) { overflow = true; break; }                                                           \
            wave_sync(); flush_trace(len); wave_sync();                                                                 \
            if (lane == 56) tp = &L.tbuf[0];                                                                            \
            room = min(TRACE_LDS, max_trace - len);                                                                     \
        }                                                                                                               \
        {   /* lane 56 holds piece 7: the entry is (this node, own observation, own score, header) */                   \
            uint4 te_ = RC;                                                                                             \
            te_.x = (uint32_t)cur_node;                                                                                 \
            *tp = te_;                                                                                                  \
        }                                                                                                               \
        tp += tinc;                                                                                                     \
        len += 1;                                                                                                       \
        room -= 1;                                                                                                      \
        const uint64_t onm = __builtin_amdgcn_ballot_w64(RC.x != 0u) & lanes_lt56;                                      \
        if (onm == 0ull) break;                       /* no children: a leaf */                                         \
        /* the next level's round of loads, on the assumption that the predicted child is the one */                   \
        PN = rl_u32(RN.x, 56);                                                                                          \
        SN = buf_ld16(stat_rs, RN.y * 16u);                                                                             \
        RNN = buf_ld16(rec_rs, PN * (TM_REC_DW * 4u) + grp16);                                                          \
        __builtin_amdgcn_sched_barrier(0);        /* keep their issue ahead of the arithmetic below */                  \
        const int visit = (int)(SC.x & vmask);                                                                          \
        const uint64_t lowmask = __builtin_amdgcn_ballot_w64(visit < low) & onm;                                        \
        uint64_t selmask;                         /* lanes of the selected group (at least its first) */                \
        if (__builtin_expect(lowmask != 0ull, 0)) {                                                                     \
            /* check_low (core.h:65-77): a uniformly drawn under-visited child, libc rand() */                          \
            uint64_t mm = lowmask & 0x0101010101010101ull;      /* one bit per group */                                 \
            const int m = __popcll(mm);                                                                                 \
            const uint32_t r = wave_rand_lds(L.misc, rng_pos, lane);                                                    \
            const int kth = (int)(r % (uint32_t)m);                                                                     \
            for (int t = 0; t < kth; ++t) mm &= mm - 1;                                                                 \
            selmask = mm;                                                                                               \
        } else {                                                                                                        \
            /* policy_clt (core.h:83-105): float arithmetic, one rounding per operation; the exploration term           \
               sqrtf(variance / (float)visit) was evaluated (same two operations) when the statistics changed.          \
               n = sum of the children's visits (empty slots read the null observation: 0 visits). */                   \
            const int n = (int)group_sum_u32((uint32_t)visit);                                                          \
            int cbits;                                                                                                  \
            if (__builtin_expect(n >= nq_size, 0)) {                                                                    \
                cbits = __builtin_amdgcn_readfirstlane(__float_as_int(norm_quantile_dev((double)n)));                   \
                nq_fallback += 1;                                                                                       \
            } else {                                                                                                    \
                TM_NQ_LOOKUP(cbits, n)                                                                                  \
            }                                                                                                           \
            const float t1 = __uint_as_float(SC.y) + __uint_as_float(RC.z);                                             \
            float val = t1 - __uint_as_float(RC.w);                                                                     \
            const float prod = __int_as_float(cbits) * __uint_as_float(SC.w);                                           \
            const float q = (val + prod) + 0.0f;         /* -0 becomes +0 (the float compare treats them as equal) */   \
            /* first-max argmax with the reference's scan semantics (max_q = q_0; i >= 1 replaces only if q_i > max_q): \
               a NaN at i >= 1 never wins, a NaN at 0 is never replaced, ties keep the lower index.  Compared as        \
               order-preserving integer keys. */                                                                        \
            const uint32_t qb = __float_as_uint(q);                                                                     \
            int key = (int)(qb ^ (((uint32_t)((int)qb >> 31)) >> 1));                                                   \
            key = (q != q) ? nan_key : key;                                                                             \
            key = (RC.x != 0u && lane < 56) ? key : (int)0x80000000;                                                    \
            const int kmax = group_max_i32(key);                                                                        \
            selmask = __builtin_amdgcn_ballot_w64(key == kmax);                                                         \
        }                                                                                                               \
        const uint32_t c = rl_u32(RC.x, __builtin_ctzll(selmask) & 56);                                                 \
        if (__builtin_expect(c != PC, 0)) {                                                                             \
            /* another child than the predicted one: remember it for the next walk through this node, fetch its         \
               record, restart the pipeline */                                                                          \
            if (lane == 56) P.rec()[(size_t)cur_node * TM_REC_DW + TM_REC_PCHILD] = c;                                  \
            RN = buf_ld16(rec_rs, c * (TM_REC_DW * 4u) + grp16);                                                        \
            PN = rl_u32(RN.x, 56);                                                                                      \
            SN = buf_ld16(stat_rs, RN.y * 16u);                                                                         \
            RNN = buf_ld16(rec_rs, PN * (TM_REC_DW * 4u) + grp16);                                                      \
            n_miss += 1;                                                                                                \
        }                                                                                                               \
        cur_node = (int)c;                                                                                              \
    }
    for (;;) {
        TM_WALK_LEVEL(r0, p0, s0, r1, p1, s1, r2)
        TM_WALK_LEVEL(r1, p1, s1, r2, p2, s0, r0)
        TM_WALK_LEVEL(r2, p2, s0, r0, p0, s1, r1)
        TM_WALK_LEVEL(r0, p0, s1, r1, p1, s0, r2)
        TM_WALK_LEVEL(r1, p1, s0, r2, p2, s1, r0)
        TM_WALK_LEVEL(r2, p2, s1, r0, p0, s0, r1)
    }
#undef TM_WALK_LEVEL
#undef TM_NQ_LOOKUP
    idx = cur_node;
    wave_sync();
    // piece 7 of the node the walk ended at = the last trace entry (on a trace overflow the node is not recorded: it is
    // treated as terminal and none of these is used)
    const uint4 cur = L.tbuf[(len - 1 - flushed) & (TRACE_LDS - 1)];
    const uint32_t rng_keep = (lane < 32) ? L.misc[lane] : 0u;     // L.misc is reused by the expansion
    hdr = cur.w;
    self_o = cur.y;
    const uint32_t self_sc = cur.z;
    wave_sync();
    flush_trace(len);
    const long long tc_sel = __builtin_readcyclecounter();
    const int leaf = idx;
    const int leaf_end = (int)((hdr >> 24) & 1u);
    int leaf_score = (int)P.game()[(size_t)leaf * GAME_DW + 14];
    if (overflow) { if (lane == 0) atomicOr(&P.gs()[TM_GS_ERR], TM_ERR_TRACE); }
    if (VANILLA && !leaf_end && !overflow) {
        // Vanilla.py:47-55: play a copy of the leaf to the end with uniformly random actions, value = final score
        uint32_t* ms = S.mt_state + (size_t)g * 625;
        for (int i = lane; i < 625; i += 64) { if (i < 624) M->mt[i] = ms[i]; else M->idx = ms[i]; }
        if (lane < GAME_DW) L.slots[7][lane] = P.game()[(size_t)leaf * GAME_DW + lane];
        wave_sync();
        if (lane == 0) {
            EngCfg cfg{S.app, S.scoring, S.randomizer};
            Piece p;
            load_fields(L.slots[7], p);
            // Vanilla.py:52 draws randint(0, 6); VanillaC.py:7 randint(0, 7) (action 7 does nothing but let gravity act)
            if (S.kind == TM_KIND_VANILLA_C)
                while (!(p.flags & 1)) play(reinterpret_cast<uint16_t*>(L.slots[7]), p, cfg, mt_randint8(*M), nullptr);
            else
                while (!(p.flags & 1)) play(reinterpret_cast<uint16_t*>(L.slots[7]), p, cfg, mt_randint7(*M), nullptr);
            L.misc[60] = (uint32_t)p.score;
        }
        wave_sync();
        leaf_score = (int)L.misc[60];
        for (int i = lane; i < 625; i += 64) ms[i] = (i < 624) ? M->mt[i] : M->idx;
        wave_sync();
    }
    // everything the expansion stage needs is in the control block: it is redone after a collection
    if (lane == 0) {
        int32_t* gs = P.gs();
        gs[TM_GS_CYC_SELECT] = (int)(tc_sel - tc_start);
        gs[TM_GS_CYC_VERIFY] = (int)(tc_ver - tc_start);
        gs[TM_GS_TRACE_LEN] = len;
        gs[TM_GS_LEAF] = leaf;
        gs[TM_GS_LEAF_END] = leaf_end | (overflow ? 1 : 0);
        gs[TM_GS_LEAF_SCORE] = leaf_score;
        gs[TM_GS_TRACE_SUM] = GSV(gsv, TM_GS_TRACE_SUM) + len;
        if (len > GSV(gsv, TM_GS_MAX_TRACE)) gs[TM_GS_MAX_TRACE] = len;
        if (nq_fallback) gs[TM_GS_N_NQ_FALLBACK] = GSV(gsv, TM_GS_N_NQ_FALLBACK) + nq_fallback;
        gs[TM_GS_SIM_STARTED] = GSV(gsv, TM_GS_SIM_STARTED) + 1;
        gs[TM_GS_N_WALK_MISS] = GSV(gsv, TM_GS_N_WALK_MISS) + n_miss;
        gs[TM_GS_FIRST_MISS] = k0;
        gs[TM_GS_PREFIX_SUM] = GSV(gsv, TM_GS_PREFIX_SUM) + k0;
    }
    if (rng_pos != rng_pos0) {
        if (lane < 32) S.rng[(size_t)g * 32 + lane] = rng_keep;
        if (lane == 0) P.gs()[TM_GS_RNG_POS] = rng_pos;
    }
    wave_front_finish<VANILLA>(S, P, L, g, lane, leaf, leaf_end | (overflow ? 1 : 0), self_o, self_sc, gsv, gc_req_word, eflags);
}

__device__ __forceinline__ bool bit_test_set(uint8_t* bm, uint32_t i) {
    uint32_t* w = reinterpret_cast<uint32_t*>(bm) + (i >> 5);
    uint32_t m = 1u << (i & 31);
    uint32_t old = atomicOr(w, m);
    return (old & m) != 0;
}
__device__ __forceinline__ bool bit_test(const uint8_t* bm, uint32_t i) {
    return (reinterpret_cast<const uint32_t*>(bm)[i >> 5] >> (i & 31)) & 1u;
}

// ---------------------------------------------------------------------------------------------------
// DistValueSim (TM_KIND_DIST): the reference's unfinished distributional agent rebuilt from its working parts
// (agents/DistValueSimOnline.py:12-94 does not import; agents/core_distributional.py:12-124 holds the kernels).
// Tree without the observation projection: obs_stat[node] = (visit, mean, variance, M2) as floats (node_stats of
// core_distributional.py; the score is in the record), node_dist[node] = `bins` atoms over [vmin, vmax).  The numerics are
// those of the oracle's restatement (oracle/dist_oracle.c orc_distpy_policy / orc_distpy_backup, pinned on a pure-Python
// run of the reference functions): float32 arrays, float64 scalars, an element op a float64 scalar done in double and
// rounded on the store.
// ---------------------------------------------------------------------------------------------------
// backup_trace_distributional (core_distributional.py:108-124): every node of the trace gets the leaf's distribution
// shifted by the reward between it and the leaf (shift_distribution :12-37) averaged into its own, and a Welford update of
// its mean / M2 / variance.  One lane per atom; a destination bin gathers its (at most a handful of) source bins in
// ascending order, which is the order the reference's scatter loop adds them in.
__device__ __forceinline__ void wave_dist_back(const tm_store& S, const GP& P, WaveLds& L, int g, int lane, int gsv) {
    const int len = GSV(gsv, TM_GS_TRACE_LEN);
    const int leaf_end = GSV(gsv, TM_GS_LEAF_END);
    const int bins = S.dist_bins;
    const double vmin = S.dist_vmin, vmax = S.dist_vmax;
    const double delta = (vmax - vmin) / bins;
    const double r = (double)GSV(gsv, TM_GS_LEAF_SCORE);               // value = leaf_game.getScore()
    float* dl = reinterpret_cast<float*>(L.slots);                     // the leaf's distribution, in LDS for the gathers
    // v_dummy (all mass in bin 0) for a finished game, else the evaluator's output (DistValueSimOnline.py:25-26,66-74)
    const float d_own = lane < bins ? (leaf_end ? (lane == 0 ? 1.0f : 0.0f) : S.eval_dist[(size_t)g * TM_DIST_ROW + lane]) : 0.0f;
    dl[lane] = d_own;
    wave_sync();
    double mean = 0;                                                   // mean_dist (:40-46), bins in ascending order
    for (int b = 0; b < bins; ++b) mean = mean + (double)dl[b] * (((double)b + 0.5) * delta);
    float* nd_base = S.node_dist + (size_t)g * P.n() * TM_DIST_ROW;
    // one node of the trace: e = its trace entry (node, node, score bits, header), sv / nd_old = its statistics and this lane's
    // atom as they were
    auto update = [&](const uint4 e, const uint4 sv, const float nd_old) {
        const int idx = (int)e.x;
        uint32_t* st = P.stat() + (size_t)idx * 4;
        float* nd = nd_base + (size_t)idx * TM_DIST_ROW;
        const float n0 = __uint_as_float(sv.x), m_old = __uint_as_float(sv.y), var_old = __uint_as_float(sv.z), m2_old = __uint_as_float(sv.w);
        const double _r = r - (double)__uint_as_float(e.z);
        const double bin_shift = _r / delta;
        const double fraction = bin_shift - floor(bin_shift);
        // shift_distribution as a gather: the sources of destination bin j are the b with lo(b) == j (weight 1 - fraction)
        // or hi(b) == j (weight fraction), lo(b) = min(int(b + bin_shift), bins - 1), hi(b) = min(lo + 1, bins - 1); for
        // j < bins - 1 they lie in [j - s - 2, j - s + 1] (s = int(bin_shift); the slack covers the rounding of b +
        // bin_shift), for the top bin in [bins - 1 - s - 2, bins - 1]
        float acc = 0.0f;
        if (lane < bins) {
            const int sft = (int)bin_shift;
            const int b_first = max(0, lane - sft - 2), b_last = lane == bins - 1 ? bins - 1 : min(bins - 1, lane - sft + 1);
            for (int b = b_first; b <= b_last; ++b) {
                int lo = (int)((double)b + bin_shift);
                if (lo >= bins) lo = bins - 1;
                const int hi = (lo + 1 >= bins) ? bins - 1 : lo + 1;
                // (negative indices - a node scoring more than the leaf - do not occur: the score never decreases along a game)
                if (lo == lane) acc = (float)((double)acc + (double)dl[b] * (1 - fraction));
                if (hi == lane) acc = (float)((double)acc + (double)dl[b] * fraction);
            }
            const float m = nd_old * n0;
            const float u = m + acc;
            nd[lane] = (float)((double)u / ((double)n0 + 1.0));
        }
        if (lane == 0) {
            const double x = mean + _r;
            const float n1 = n0 + 1.0f;
            const double d1 = x - (double)m_old;
            const float v = (float)((double)m_old + d1 / (double)n1);
            const double d2 = x - (double)v;
            const float m2 = (float)((double)m2_old + d1 * d2);
            const float var = n1 > 1.0f ? (float)((double)m2 / ((double)n1 - 1.0)) : var_old;
            *reinterpret_cast<uint4*>(st) = make_uint4(__float_as_uint(n1), __float_as_uint(v), __float_as_uint(var), __float_as_uint(m2));
        }
    };
    if (S.app == 1) {
        // With a descent at every action a path never visits a node twice, so the nodes' updates are independent of each
        // other: the loads of DB_CHUNK nodes (entries, then statistics + distribution rows) are in flight together - two
        // memory round trips per chunk instead of two per node (the backup was 45 of the mean wave's 104 kcycles).
        constexpr int DB_CHUNK = 8;
        for (int base = 0; base < len; base += DB_CHUNK) {
            uint4 e[DB_CHUNK], sv[DB_CHUNK];
            float ndo[DB_CHUNK];
#pragma unroll
            for (int k = 0; k < DB_CHUNK; ++k)
                e[k] = base + k < len ? reinterpret_cast<const uint4*>(P.trace())[base + k] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int k = 0; k < DB_CHUNK; ++k) {
                sv[k] = *reinterpret_cast<const uint4*>(P.stat() + (size_t)e[k].x * 4);        // (entry 0 = node 0: never a real node, harmless to read)
                ndo[k] = lane < bins ? nd_base[(size_t)e[k].x * TM_DIST_ROW + lane] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < DB_CHUNK; ++k)
                if (base + k < len) update(e[k], sv[k], ndo[k]);
        }
    } else {
        for (int i = 0; i < len; ++i) {
            const uint4 e = reinterpret_cast<const uint4*>(P.trace())[i];   // (node, node, score bits, header)
            const uint4 sv = *reinterpret_cast<const uint4*>(P.stat() + (size_t)e.x * 4);
            const float nd_old = lane < bins ? (nd_base + (size_t)e.x * TM_DIST_ROW)[lane] : 0.0f;
            update(e, sv, nd_old);
        }
    }
    wave_sync();
    if (lane == 0) { P.gs()[TM_GS_PENDING] = 0; P.gs()[TM_GS_N_SIMS] = GSV(gsv, TM_GS_N_SIMS) + 1; }
}

// select_trace_distributional (core_distributional.py:81-104): under-visited children first (check_low, libc rand()),
// else policy_dist (:66-79): q = (mean + score - own score) + norm_quantile(n) * sqrt(variance / (visit + 1e-3)), first
// argmax (np.argmax: the first NaN wins).  One 8-lane group per unique child (the record's pieces), one dependent pair of
// loads per level (record -> children's statistics).
template <bool VANILLA>
__device__ __forceinline__ void wave_dist_front(const tm_store& S, const GP& P, WaveLds& L, int g, int lane, int gsv, int gc_req_word, int eflags) {
    const long long tc_start = __builtin_readcyclecounter();
    if (lane < 32) L.misc[lane] = S.rng[(size_t)g * 32 + lane];
    wave_sync();
    int rng_pos = GSV(gsv, TM_GS_RNG_POS);
    const int rng_pos0 = rng_pos;
    int idx = GSV(gsv, TM_GS_ROOT);
    int len = 0, flushed = 0;
    bool overflow = false;
    const int grp = lane >> 3;
    const uint32_t grp16 = (uint32_t)grp * 16u;
    const __amdgpu_buffer_rsrc_t rec_rs = make_rsrc(P.rec(), P.n() * (TM_REC_DW * 4u));
    const __amdgpu_buffer_rsrc_t stat_rs = make_rsrc(P.stat(), P.n() * 16u);
    const float lowf = (float)S.low;
    const double* nqd = S.nq_table_d;
    int nq_fallback = 0;
    uint4 cur = make_uint4(0, 0, 0, 0);
    // Every node remembers the child its last walk took (piece 7, word 0, as the observation walk of wave_sim_front does): that
    // child's record is requested together with the children's statistics, so a level whose selection repeats costs one memory
    // round trip instead of two; another selection rewrites the prediction.  Results do not depend on predictions.
    uint4 rc_pred = make_uint4(0, 0, 0, 0);
    uint32_t pred = 0xFFFFFFFFu;            // the node rc_pred is the record of (none yet)
    for (;;) {
        const uint4 rc = (uint32_t)idx == pred ? rc_pred : buf_ld16(rec_rs, (uint32_t)idx * (TM_REC_DW * 4u) + grp16);       // group j: piece j of the node
        if (len - flushed == TRACE_LDS) {
            if (len >= S.max_trace) { overflow = true; break; }
            wave_sync();
            for (int i = lane; i < TRACE_LDS; i += 64) reinterpret_cast<uint4*>(P.trace())[flushed + i] = L.tbuf[i];
            wave_sync();
            flushed = len;
        }
        if (len >= S.max_trace) { overflow = true; break; }
        if (lane == 56) { cur = rc; L.tbuf[len - flushed] = make_uint4((uint32_t)idx, rc.y, rc.z, rc.w); }
        len += 1;
        const bool exists = rc.x != 0u && lane < 56;
        const uint64_t onm = __builtin_amdgcn_ballot_w64(exists);
        if (onm == 0ull) break;                                  // no children: a leaf
        const uint4 sc = buf_ld16(stat_rs, (lane < 56 ? rc.x : 0u) * 16u);                 // the child NODE's statistics
        pred = rl_u32(rc.x, 56);
        if (pred != 0u) rc_pred = buf_ld16(rec_rs, pred * (TM_REC_DW * 4u) + grp16);
        else pred = 0xFFFFFFFFu;
        const float visit = exists ? __uint_as_float(sc.x) : 0.0f;
        const uint64_t lowmask = __builtin_amdgcn_ballot_w64(exists && visit < lowf);
        uint32_t c;
        if (lowmask != 0ull) {
            uint64_t mm = lowmask & 0x0101010101010101ull;       // one bit per group = per unique child, in slot order
            const int m = __popcll(mm);
            const uint32_t rr = wave_rand_lds(L.misc, rng_pos, lane);
            const int kth = (int)(rr % (uint32_t)m);
            for (int t = 0; t < kth; ++t) mm &= mm - 1;
            c = rl_u32(rc.x, __builtin_ctzll(mm) & 56);
        } else {
            const int n = (int)group_sum_u32((uint32_t)(int)visit);                        // visits are whole numbers
            double coeff;
            if (n < S.nq_size) coeff = nqd[n];
            else { const double alpha = 1 - 1 / (double)n; coeff = 10 * log(1 - log(-log(alpha) / log(2.0)) / log(22.0)) / log(41.0); nq_fallback += 1; }
            const float t1 = __uint_as_float(sc.y) + __uint_as_float(rc.z);                // mean + score, float
            const float s0 = (float)((double)t1 - (double)__uint_as_float(rc.w));         // - curr_reward (the node's own score)
            const float s1 = (float)((double)__uint_as_float(sc.z) / ((double)visit + 1e-3));
            const double q = (double)s0 + coeff * (double)sqrtf(s1);
            // np.argmax over the children in slot order: every lane runs the seven-step scan on the groups' values
            int best = 0;
            double best_q = 0;
            bool done = false;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const double qj = __longlong_as_double((long long)rl_u64((uint64_t)__double_as_longlong(q), 8 * j));
                const bool ex = (onm >> (8 * j)) & 1ull;
                if (!ex || done) continue;
                if (j == 0) { best_q = qj; continue; }
                if (best_q != best_q) { done = true; continue; }
                if (qj != qj || qj > best_q) { best_q = qj; best = j; }
            }
            c = rl_u32(rc.x, 8 * best);
        }
        if (c != pred && lane == 0) P.rec()[(size_t)idx * TM_REC_DW + TM_REC_PCHILD] = c;
        idx = (int)c;
    }
    wave_sync();
    cur = make_uint4(rl_u32(cur.x, 56), rl_u32(cur.y, 56), rl_u32(cur.z, 56), rl_u32(cur.w, 56));
    const uint32_t rng_keep = (lane < 32) ? L.misc[lane] : 0u;     // L.misc is reused by the expansion
    wave_sync();
    for (int i = flushed + lane; i < len; i += 64) reinterpret_cast<uint4*>(P.trace())[i] = L.tbuf[i - flushed];
    const long long tc_sel = __builtin_readcyclecounter();
    const int leaf = idx;
    const int leaf_end = (int)((cur.w >> 24) & 1u) | (overflow ? 1 : 0);
    const int leaf_score = (int)P.game()[(size_t)leaf * GAME_DW + 14];
    if (overflow) { if (lane == 0) atomicOr(&P.gs()[TM_GS_ERR], TM_ERR_TRACE); }
    if (lane == 0) {
        int32_t* gs = P.gs();
        gs[TM_GS_CYC_SELECT] = (int)(tc_sel - tc_start);
        gs[TM_GS_TRACE_LEN] = len;
        gs[TM_GS_LEAF] = leaf;
        gs[TM_GS_LEAF_END] = leaf_end;
        gs[TM_GS_LEAF_SCORE] = leaf_score;
        gs[TM_GS_TRACE_SUM] = GSV(gsv, TM_GS_TRACE_SUM) + len;
        if (len > GSV(gsv, TM_GS_MAX_TRACE)) gs[TM_GS_MAX_TRACE] = len;
        if (nq_fallback) gs[TM_GS_N_NQ_FALLBACK] = GSV(gsv, TM_GS_N_NQ_FALLBACK) + nq_fallback;
        gs[TM_GS_SIM_STARTED] = GSV(gsv, TM_GS_SIM_STARTED) + 1;
    }
    if (rng_pos != rng_pos0) {
        if (lane < 32) S.rng[(size_t)g * 32 + lane] = rng_keep;
        if (lane == 0) P.gs()[TM_GS_RNG_POS] = rng_pos;
    }
    wave_front_finish<VANILLA>(S, P, L, g, lane, leaf, leaf_end, cur.y, cur.z, gsv, gc_req_word, eflags);
}

// ---------------------------------------------------------------------------------------------------
// GC (agents/agent.py:206-257, ValueSim.py:101-159): reachable set from the root, free lists rebuilt in
// ascending order, freed observations cleared, replay tuples harvested, tables rebuilt.
//
// A collection is the work of a GROUP of T threads: a whole 256-thread workgroup (the collector workgroups that
// k_sim_step carries beside its simulation workgroups; tm_tree_remove_nodes), or one wave where a kernel's waves own
// different games (update_root's exhausting pop, the single-call entry points).  It is RESUMABLE: it runs in slices up
// to a deadline (shader clock; none: to completion), its progress lives in the game's control block (TM_GS_GC_*), and
// a game that is collecting simply does not simulate - the launch is never held up by the sweep of a 100 000-entry
// pool.  The simulation that hit the empty pool is suspended in its expansion (TM_GS_PENDING == 2) and redone when the
// collection is complete; the order of events inside the game is exactly the reference's.
//
// Hand-over between a game's simulation wave and a collector workgroup happens at kernel boundaries only (different
// CUs, different L2s: nothing written in a launch is read by the other side in the same launch).  The phase word
// carries the launch number for that: a request is (launch << 4) | 1 and is picked up by the collectors of LATER
// launches; a finished collection leaves (launch << 4) | 7 and the game's wave resumes in a LATER launch.
// ---------------------------------------------------------------------------------------------------
// Collector workgroups of a k_sim_step launch: half for the bounded steps, half for the marking (measured, r03: with a quarter for
// the marking a collection takes 35 launches instead of 23).  128 since r05: under a TRAINED value net every simulation expands a
// node, the pools fill twice as fast and the trees that survive a move are larger - 64 collectors kept a game waiting 78 launches
// per collection (161 ms per move in the steady state), 128 keep it 17 (103 ms); 192 or another split: no further gain
// (profiles/r05_gc_sweep_trained_net.json).  Under the random-init net of the headline window 128 cost 0.5-2 %.
constexpr int GC_BLOCKS_DEFAULT = 128;
constexpr int GC_BLOCKS_MAX = 128;      // (tm_store::gc_collectors; the bounded half keeps three counts per workgroup in TM_GC_PART_DW words)
static_assert(3 * (GC_BLOCKS_MAX / 2) <= TM_GC_PART_DW, "three counts per bounded workgroup (half of the collectors)");
__host__ __device__ inline int gc_blocks(const tm_store& S) {
    const int sim_blocks = (S.n_games + WPB - 1) / WPB;
    int want = S.gc_collectors > 0 ? S.gc_collectors : GC_BLOCKS_DEFAULT;
    if (want > GC_BLOCKS_MAX) want = GC_BLOCKS_MAX;
    return sim_blocks < want ? sim_blocks : want;
}

template <int T> struct Grp {
    static constexpr int NW = T / 64;
    // all stores of the group so far are visible to all of its threads (same CU: workgroup scope)
    static __device__ __forceinline__ void sync() {
        if constexpr (T == 64) { __threadfence_block(); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    }
    // exclusive prefix sum of v over the group's threads in thread order; total to every thread.  sm: NW + 1 ints
    static __device__ __forceinline__ int exscan(int v, int tid, int* sm, int& total) {
        const int lane = tid & 63;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if constexpr (T == 64) {
            total = __shfl(incl, 63, 64);
            return incl - v;
        } else {
            const int w = tid >> 6;
            __syncthreads();               // the previous use of sm is over
            if (lane == 63) sm[w] = incl;
            __syncthreads();
            int off = 0, tot = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) { const int x = sm[i]; off += i < w ? x : 0; tot += x; }
            total = tot;
            return off + incl - v;
        }
    }
    // thread 0's value to every thread
    static __device__ __forceinline__ int bcast(int v, int tid, int* sm) {
        if constexpr (T == 64) return __shfl(v, 0, 64);
        else {
            __syncthreads();
            if (tid == 0) sm[NW] = v;
            __syncthreads();
            return sm[NW];
        }
    }
};

// One group of T threads, the whole collection, now (update_root's exhausting pop, the single-call entry points,
// tm_tree_remove_nodes).  sm: Grp<T>::NW + 1 ints of LDS (unused for T == 64).
// The online leg of the distributional agent: DistValueSimOnline.store_nodes (agents/DistValueSimOnline.py:116-141 - commented
// out in the reference, rebuilt as SURVEY 8(f)2 asks): a node that a collection frees becomes a training tuple when it has at
// least min_visits visits and every one of its seven children has been visited (a missing child is node 0, whose visit count
// is 0); the tuple = (its board, its distribution, its visit count).  The decision reads the children's statistics, which the
// same collection may be clearing: it is taken where nothing is cleared yet (the count step / a pass of its own) and left in
// bit 26 of the node's record header; the write step copies the nodes that carry it, in index order.
constexpr uint32_t HDR_HARVEST = 1u << 26;
__device__ __forceinline__ bool dist_keep(const tm_store& S, const GP& P, int idx) {
    const float v = __uint_as_float(P.stat()[(size_t)idx * 4]);
    if (!(v >= (float)S.min_visits_to_store)) return false;
    const uint4* kd = reinterpret_cast<const uint4*>(P.kids() + (size_t)idx * TM_KIDS_DW);
    const uint4 k0 = kd[0], k1 = kd[1];
    const uint32_t c[7] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z};
    float cv[7];
#pragma unroll
    for (int a = 0; a < 7; ++a) cv[a] = __uint_as_float(P.stat()[(size_t)c[a] * 4]);
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 7; ++a) ok = ok && c[a] != 0u && cv[a] >= 1.0f;
    return ok;
}
// marks the freed nodes of bitmap word wi that are training tuples; returns how many
__device__ __forceinline__ int dist_mark_word(const tm_store& S, const GP& P, uint32_t freed, int wi, int low_node) {
    int n = 0;
    if (wi * 32 + 31 < low_node) return 0;
    for (uint32_t bits = freed; bits; bits &= bits - 1) {
        const int i = wi * 32 + (__ffs(bits) - 1);
        if (i < low_node || i == 0) continue;
        if (dist_keep(S, P, i)) {
            uint32_t* h = P.rec() + (size_t)i * TM_REC_DW + TM_REC_HDR;
            *h = *h | HDR_HARVEST;
            n += 1;
        }
    }
    return n;
}
__device__ __forceinline__ uint32_t dist_marked(const GP& P, uint32_t freed, int wi, int low_node) {
    uint32_t keep = 0;
    if (wi * 32 + 31 < low_node) return 0u;
    for (uint32_t bits = freed; bits; bits &= bits - 1) {
        const int b = __ffs(bits) - 1, i = wi * 32 + b;
        if (i >= low_node && (P.rec()[(size_t)i * TM_REC_DW + TM_REC_HDR] & HDR_HARVEST)) keep |= 1u << b;
    }
    return keep;
}
__device__ __forceinline__ void dist_harvest_store(const tm_store& S, const GP& P, int g, int idx, int kpos) {
    uint32_t gm[GAME_DW], ok[OBS_DW];
    const uint4* src = reinterpret_cast<const uint4*>(P.game() + (size_t)idx * GAME_DW);
#pragma unroll
    for (int t = 0; t < 4; ++t) { const uint4 q = src[t]; gm[4 * t] = q.x; gm[4 * t + 1] = q.y; gm[4 * t + 2] = q.z; gm[4 * t + 3] = q.w; }
    pack_obs(gm, ok);
    uint4* dk = reinterpret_cast<uint4*>(S.replay_obs + ((size_t)g * S.replay_cap + kpos) * TM_OBS_DW);
    dk[0] = make_uint4(ok[0], ok[1], ok[2], ok[3]); dk[1] = make_uint4(ok[4], ok[5], ok[6], ok[7]); dk[2] = make_uint4(ok[8], ok[9], ok[10], ok[11]);
    float* ds = S.replay_stat + ((size_t)g * S.replay_cap + kpos) * 4;
    ds[0] = 0.f; ds[1] = 0.f; ds[2] = __uint_as_float(P.stat()[(size_t)idx * 4]); ds[3] = 0.f;      // weight = the visit count
    const uint4* sd = reinterpret_cast<const uint4*>(S.node_dist + ((size_t)g * P.n() + (size_t)idx) * TM_DIST_ROW);
    uint4* dd = reinterpret_cast<uint4*>(S.replay_dist + ((size_t)g * S.replay_cap + kpos) * TM_DIST_ROW);
#pragma unroll
    for (int t = 0; t < TM_DIST_ROW / 4; ++t) dd[t] = sd[t];
}

template <int T>
__device__ __forceinline__ void gc_collect(const tm_store& S, const GP& P, int g, int tid, int* sm) {
    typedef Grp<T> G_;
    constexpr int U = 4;      // queue entries per thread and marking round: their loads and atomics are in flight together
    constexpr int UR = 2;     // keys per thread and re-insertion round
    const int lane = tid & 63;
    const long long gc_t0 = __builtin_readcyclecounter();
    int32_t* gs = P.gs();
    const int N = S.max_nodes;
    const size_t bm_bytes = (((size_t)N + 7) / 8 + 15) & ~(size_t)15;
    uint8_t* nmark = S.gc_mark + (size_t)g * 2 * bm_bytes;
    uint8_t* omark = nmark + bm_bytes;
    uint32_t* nmw = reinterpret_cast<uint32_t*>(nmark);
    uint32_t* omw = reinterpret_cast<uint32_t*>(omark);
    int32_t* queue = S.gc_queue + (size_t)g * N;
    const uint32_t mask = (uint32_t)S.table_cap - 1u;
    int tail = 1, nfree = 0, onfree = 0;
    {
        for (size_t i = tid; i < 2 * bm_bytes / 4; i += T) nmw[i] = 0;      // both bitmaps
        G_::sync();
        // breadth-first marking; index 0 is followed like any other child (core.h:41-45), so it stays occupied
        if (tid == 0) { queue[0] = gs[TM_GS_ROOT]; bit_test_set(nmark, (uint32_t)queue[0]); }
        G_::sync();
    }
    {
        int head = 0;
        while (head < tail) {
            const int stop = min(tail, head + T * U);
            uint32_t ch[U][7];
            uint32_t ob[U];
            bool val[U];
            // all loads of the round, then all test-and-set atomics (their latencies overlap), then the results
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = head + u * T + tid;
                val[u] = i < stop;
                const int node = val[u] ? queue[i] : 0;
                const uint4 k0 = *reinterpret_cast<const uint4*>(P.kids() + (size_t)node * TM_KIDS_DW);
                const uint4 k1 = *reinterpret_cast<const uint4*>(P.kids() + (size_t)node * TM_KIDS_DW + 4);
                ob[u] = P.rec()[(size_t)node * TM_REC_DW + TM_REC_OBS];
                ch[u][0] = k0.x; ch[u][1] = k0.y; ch[u][2] = k0.z; ch[u][3] = k0.w; ch[u][4] = k1.x; ch[u][5] = k1.y; ch[u][6] = k1.z;
            }
            uint32_t fresh[U];      // bit a: child a was not marked before this round
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // several actions of a node often lead to the same child (moves against a wall): one atomic per distinct child
                uint32_t skip = val[u] ? 0u : 0x7Fu;
#pragma unroll
                for (int a = 1; a < 7; ++a)
#pragma unroll
                    for (int b = 0; b < a; ++b)
                        if (ch[u][a] == ch[u][b]) skip |= 1u << a;
                uint32_t old[7];
#pragma unroll
                for (int a = 0; a < 7; ++a)
                    old[a] = ((skip >> a) & 1u) ? 0xFFFFFFFFu : atomicOr(nmw + (ch[u][a] >> 5), 1u << (ch[u][a] & 31));
                if (val[u]) atomicOr(omw + (ob[u] >> 5), 1u << (ob[u] & 31));
                uint32_t f = 0;
#pragma unroll
                for (int a = 0; a < 7; ++a)
                    if (!((old[a] >> (ch[u][a] & 31)) & 1u)) f |= 1u << a;
                fresh[u] = f;
            }
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) cnt += __popc(fresh[u]);
            int total;
            int off = tail + G_::exscan(cnt, tid, sm, total);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int a = 0; a < 7; ++a)
                    if ((fresh[u] >> a) & 1u) queue[off++] = (int)ch[u][a];
            head = stop;
            tail += total;
            G_::sync();        // the queue is produced and consumed by this group only (same CU)
        }
        // the root of an unexpanded tree still reaches node 0 through its zero child row
    }
    {
        // both tables are rebuilt from what is kept: clear them first (nothing below reads them).  16-byte stores: two
        // 8-byte entries each (table_cap is a power of two, the per-game tables are 16-byte aligned)
        uint4* nt4 = reinterpret_cast<uint4*>(P.ntab());
        uint4* ot4 = reinterpret_cast<uint4*>(P.otab());
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        const int n4 = S.table_cap / 2;
        for (int i = tid; i < n4; i += T) { nt4[i] = z4; ot4[i] = z4; }
        G_::sync();
    }
    // Sweep over the pool: each thread takes one 32-bit word of a bitmap = 32 consecutive indices per pass.  Free lists
    // come out ascending (agents/agent.py:211-212,221-222): thread order is index order, positions from a group prefix sum.
    // Freed node records are not cleared here: new_node initialises a slot completely when it hands it out (the reference
    // zeroes at GC, agents/agent.py:227-244; nothing reads a free slot in between).
    {
        const int n_words = (N + 31) / 32;
        const int low_obs = gs[TM_GS_LOW_OBS], low_node = gs[TM_GS_LOW_NODE];
        const bool dharvest = S.online && S.replay_cap > 0 && S.kind == TM_KIND_DIST && S.replay_dist;
        const bool harvest = S.online && S.replay_cap > 0 && S.kind != TM_KIND_DIST;
        int m = (harvest || dharvest) ? S.replay_count[g] : 0;
        if (dharvest) {      // the tuples are chosen before anything is cleared (dist_keep reads the children's statistics)
            for (int wi = tid; wi < n_words; wi += T) {
                const int rem = N - wi * 32;
                dist_mark_word(S, P, ~nmw[wi] & (rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u)), wi, low_node);
            }
            G_::sync();
        }
        for (int wbase = 0; wbase < n_words; wbase += T) {
            const int wi = wbase + tid;
            uint32_t valid = 0;
            if (wi < n_words) { const int rem = N - wi * 32; valid = rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u); }
            // ---- nodes ----
            const uint32_t fr = (wi < n_words) ? (~nmw[wi] & valid) : 0u;
            if (dharvest) {
                const uint32_t keepmask = dist_marked(P, fr, wi, low_node);
                int ktotal;
                int kpos = m + G_::exscan(__popc(keepmask), tid, sm, ktotal);
                for (uint32_t bits = keepmask; bits; bits &= bits - 1) {
                    if (kpos < S.replay_cap) dist_harvest_store(S, P, g, wi * 32 + (__ffs(bits) - 1), kpos);
                    kpos += 1;
                }
                if (tid == 0 && m + ktotal > S.replay_cap) gs[TM_GS_N_DROPPED] += m + ktotal - S.replay_cap;
                m = min(S.replay_cap, m + ktotal);
                G_::sync();      // the copies have read what the loops below clear
            }
            int total;
            int pos = nfree + G_::exscan(__popc(fr), tid, sm, total);
            for (uint32_t bits = fr; bits; bits &= bits - 1) {
                const int i = wi * 32 + (__ffs(bits) - 1);
                P.fnode()[pos++] = i;
                if (i >= low_node) {        // a free slot holds no child indices and no observation (see the collectors' write step)
                    const uint4 z4 = make_uint4(0, 0, 0, 0);
                    uint4* kd = reinterpret_cast<uint4*>(P.kids() + (size_t)i * TM_KIDS_DW);
                    kd[0] = z4; kd[1] = z4;
                    reinterpret_cast<uint4*>(P.rec() + (size_t)i * TM_REC_DW)[7] = z4;
                }
            }
            nfree += total;
            // ---- observations ----
            const uint32_t ofr = (wi < n_words) ? (~omw[wi] & valid) : 0u;
            int ototal;
            int opos = onfree + G_::exscan(__popc(ofr), tid, sm, ototal);
            if (harvest) {
                // store_nodes (ValueSim.py:122-159): freed observations with enough visits that are not terminal
                uint32_t keepmask = 0;
                for (uint32_t bits = ofr; bits;) {
                    int obx[4];
                    uint4 st4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        obx[u] = bits ? (__ffs(bits) - 1) : -1;
                        bits &= bits ? bits - 1 : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int o = wi * 32 + (obx[u] < 0 ? 0 : obx[u]);
                        st4[u] = (obx[u] >= 0 && o >= low_obs) ? *reinterpret_cast<const uint4*>(P.stat() + (size_t)o * 4) : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (obx[u] >= 0 && (int)st4[u].x >= S.min_visits_to_store && !(st4[u].x >> 31)) keepmask |= 1u << obx[u];
                }
                int ktotal;
                int kpos = m + G_::exscan(__popc(keepmask), tid, sm, ktotal);
                for (uint32_t bits = keepmask; bits; bits &= bits - 1) {
                    const int o = wi * 32 + (__ffs(bits) - 1);
                    if (kpos < S.replay_cap) {
                        const uint4 st = *reinterpret_cast<const uint4*>(P.stat() + (size_t)o * 4);
                        const uint4* sk = reinterpret_cast<const uint4*>(P.okey() + (size_t)o * TM_OBS_DW);
                        uint4* dk = reinterpret_cast<uint4*>(S.replay_obs + ((size_t)g * S.replay_cap + kpos) * TM_OBS_DW);
                        dk[0] = sk[0]; dk[1] = sk[1]; dk[2] = sk[2];
                        float* ds = S.replay_stat + ((size_t)g * S.replay_cap + kpos) * 4;
                        ds[0] = __uint_as_float(st.y); ds[1] = __uint_as_float(st.z); ds[2] = (float)(int)st.x; ds[3] = 0.f;
                    }
                    kpos += 1;
                }
                // never silently (the reference keeps all up to memory_size)
                if (tid == 0 && m + ktotal > S.replay_cap) gs[TM_GS_N_DROPPED] += m + ktotal - S.replay_cap;
                m = min(S.replay_cap, m + ktotal);
            }
            // Of a freed observation only the statistics are cleared (16 B): a visit count of 0 is what keeps a slot that
            // stays free from being harvested again at the next GC.  Fully free words: one contiguous 512-byte store by
            // the word's wave.  (The harvest above has read its statistics: a wave's loads are complete before its stores
            // are issued only through the data dependence - hence the explicit group barrier.)
            if (harvest) G_::sync();
            const bool ofull = (wi < n_words) && valid == 0xFFFFFFFFu && ofr == 0xFFFFFFFFu && (wi * 32 >= low_obs);
            for (uint64_t fm = __ballot(ofull); fm; fm &= fm - 1) {
                const int w0 = (wbase + (tid & ~63) + (__ffsll((long long)fm) - 1)) * 32;
                if (lane < 32) reinterpret_cast<uint4*>(P.stat() + (size_t)w0 * 4)[lane] = make_uint4(0, 0, 0, 0);
            }
            for (uint32_t bits = ofr; bits; bits &= bits - 1) {
                const int o = wi * 32 + (__ffs(bits) - 1);
                P.fobs()[opos++] = o;
                if (o >= low_obs && !ofull) *reinterpret_cast<uint4*>(P.stat() + (size_t)o * 4) = make_uint4(0, 0, 0, 0);
            }
            onfree += ototal;
            if (harvest) { G_::sync(); if (tid == 0) S.replay_count[g] = m; }
        }
        if (dharvest && tid == 0) S.replay_count[g] = m;
        G_::sync();
    }
    {
        // re-insert the kept nodes (the queue lists them): threads claim empty slots with a 64-bit compare-and-swap (no
        // deletions happen concurrently, so linear probing stays consistent; placement order does not affect lookups)
        for (int base = 0; base < tail; base += T * UR) {
            uint4 key[UR][4];
            int idx[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int q = base + u * T + tid;
                idx[u] = (q < tail) ? queue[q] : 0;
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const uint4* src = reinterpret_cast<const uint4*>(P.game() + (size_t)idx[u] * GAME_DW);
#pragma unroll
                for (int t = 0; t < 4; ++t) key[u][t] = src[t];
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                if (idx[u] == 0) continue;
                uint32_t kw[GAME_DW];
#pragma unroll
                for (int t = 0; t < 4; ++t) { kw[4*t] = key[u][t].x; kw[4*t+1] = key[u][t].y; kw[4*t+2] = key[u][t].z; kw[4*t+3] = key[u][t].w; }
                const uint64_t h = hash_game(kw);
                uint32_t sl = (uint32_t)h & mask;
                const unsigned long long ent = ((h >> 32) << 32) | (uint32_t)idx[u];
                while (atomicCAS(reinterpret_cast<unsigned long long*>(&P.ntab()[sl]), 0ull, ent) != 0ull) sl = (sl + 1) & mask;
            }
        }
    }
    {
        // the kept observations, by index (the bitmap says which), back into their (cleared) table
        for (int base = 0; base < (S.kind == TM_KIND_DIST ? 0 : N); base += T * UR) {
            uint4 key[UR][3];
            bool kept[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int o = base + u * T + tid;
                kept[u] = o > 0 && o < N && bit_test(omark, (uint32_t)o);
                const uint4* src = reinterpret_cast<const uint4*>(P.okey() + (size_t)(kept[u] ? o : 0) * OBS_DW);
#pragma unroll
                for (int t = 0; t < 3; ++t) key[u][t] = src[t];
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                if (!kept[u]) continue;
                const int o = base + u * T + tid;
                uint32_t kw[OBS_DW];
#pragma unroll
                for (int t = 0; t < 3; ++t) { kw[4*t] = key[u][t].x; kw[4*t+1] = key[u][t].y; kw[4*t+2] = key[u][t].z; kw[4*t+3] = key[u][t].w; }
                const uint64_t h = hash_obs(kw);
                uint32_t sl = (uint32_t)h & mask;
                const unsigned long long ent = ((h >> 32) << 32) | (uint32_t)o;
                while (atomicCAS(reinterpret_cast<unsigned long long*>(&P.otab()[sl]), 0ull, ent) != 0ull) sl = (sl + 1) & mask;
            }
        }
    }
    G_::sync();
    if (tid == 0) {
        gs[TM_GS_NFREE_NODE] = nfree;
        gs[TM_GS_NFREE_OBS] = onfree;
        gs[TM_GS_N_GC] += 1;
        gs[TM_GS_CYC_TAIL] = (int)(((long long)__builtin_readcyclecounter() - gc_t0) >> 4);   // last GC, units of 16 cycles
        gs[TM_GS_CYC_TAIL + 1] = tail;                                                // reachable nodes at the last GC
        gs[TM_GS_GC_PHASE] = 0;
        gc_active_clear(P);
        gs[TM_GS_GC_SLICES] += 1;
    }
    G_::sync();
}
// to completion, now, by the game's own wave (update_root's exhausting pop: once per move at most, and almost never; the
// single-call entry points)
// a speculative marking is dropped whenever the tree changes without its barrier watching (update_root, the single calls)
__device__ __forceinline__ void gc_drop_speculative(const GP& P, int lane) {
    const int ph = P.gs()[TM_GS_GC_PHASE] & 15;
    wave_sync();
    if (lane == 0 && (ph == GC_SPEC_REQ || ph == GC_SPEC_MARK)) { P.gs()[TM_GS_GC_PHASE] = 0; gc_active_clear(P); }
    if (lane == 0) P.gs()[TM_GS_GC_IN_MOVE] = 0;      // (and what a collection left behind no longer says what is reachable)
    wave_sync();
}
__device__ __forceinline__ void gc_wave(const tm_store& S, const GP& P, int g, int lane) {
    if (lane == 0) { P.gs()[TM_GS_GC_PHASE] = GC_REQ; gc_active_set(P); }
    __threadfence_block();
    gc_collect<64>(S, P, g, lane, nullptr);
}

// ---------------------------------------------------------------------------------------------------
// The same collection as the work of ALL collector workgroups of a k_sim_step launch (one CU can not sweep a
// 100 000-entry pool quickly: a round trip costs over a microsecond beside 4096 walking waves).  Every collecting game
// goes through five steps, one launch each except the marking; in each, collector workgroup c of n does share c of the
// work, then ARRIVES (one device-scope atomic per game and workgroup); the last to arrive moves the game to the next
// step.  The kernel boundary is the barrier between steps: nothing a workgroup writes is read by another in the same
// launch, except through those atomics and the mark bitmaps (device-scope atomics as well).
//   1 init    the two tables cleared by the bounded workgroups - and the marking begun in the same launch by the game's
//             marking workgroup (a speculative marking: only the bitmaps and the root, by workgroup g % n: the marking starts
//             when the game's write barrier is up, one launch later)
//   2 mark    ONE marking workgroup per game and launch, both bitmaps in its LDS: gc_sweep_mark below.  Complete when it
//             leaves no chunk flagged.
//   3 count   free nodes / free observations / harvested tuples per share of the index range
//   4 write   ascending free lists (agents/agent.py:211-212,221-222), replay tuples in index order, freed statistics
//             cleared - at the offsets the counts give
//   5, 6      kept nodes, then kept observations (the bitmaps say which), back into their tables (compare-and-swap)
// All workgroups see the same games in the same steps: the control blocks they read were written in earlier launches.
// A collection in progress must be continued by launches over the same range of games (same n).
// ---------------------------------------------------------------------------------------------------
constexpr int GCP_INIT = GC_REQ, GCP_MARK = 2, GCP_COUNT = 3, GCP_WRITE = 4;      // (5, 6: the re-insertions' own steps until r06)
constexpr int GCP_LAST = 6;
constexpr int GC_LIST_MAX = 256, GC_LIST_WAIT = 64;     // collecting games looked after per launch (the others wait): all, and those that are waiting for their collection
constexpr int GC_COST_MAX = 13;      // per launch: cost units of the steps whose shares are done without looking at the clock
                                     // (init 1, count + nodes 6, write + observations 7: about 5 microseconds a unit; 13 = one of each)
constexpr int GC_WL_WORDS = 16;       // the marker's work list: 512 units (a unit = a block for pools up to 131 072 nodes)
struct GcLds {
    int scan[8];                     // Grp<256> scratch
    int n_list;
    int list_g[GC_LIST_MAX], list_ph[GC_LIST_MAX];       // the collecting games and their phase words as of the start of the launch
    unsigned char list_step[GC_LIST_MAX];               // the launch's plan: the step performed for the game,
    short list_part[GC_LIST_MAX], list_parts[GC_LIST_MAX], list_share[GC_LIST_MAX];     // this workgroup's share (part of parts; parts 0: not in this launch)
    unsigned char list_mark[GC_LIST_MAX];               // ... and whether this workgroup is the game's marker in this launch
    unsigned char list_mpart[GC_LIST_MAX], list_mparts[GC_LIST_MAX];     // ... with which share of the game's index range (part of parts)
    int n_order;
    union {
        int hist[256];                                  // (building the list) more than GC_LIST_WAIT games waiting: how many have waited how long
        struct { short order[GC_LIST_MAX];              // (the plan) the games this workgroup works on, in order
                 short mk_list[GC_LIST_MAX]; };         // the games that are marked in this launch, in the order of the plan
    };
    int age[GC_LIST_WAIT]; short by_age[GC_LIST_WAIT];  // the waiting games: launches since the request, and sorted by that
    uint32_t dirty[TM_GC_FLAG_WORDS];                   // chunk flags on their way between the control block and the work list
    uint32_t rmask[TM_GC_FLAG_WORDS];                   // which chunks this workgroup's share of the game under work is
    uint32_t fsend[WPB][TM_GC_FLAG_WORDS];              // per wave: chunks of other shares to which it has sent nodes
    uint32_t wl[GC_WL_WORDS];                           // the marker's work list: flagged units (blocks) of the share
    int busy, quit;                                     // waves of the workgroup that hold a unit; nothing will come any more
    int zero;                                           // a row with a zero entry has been looked at (node 0 is reachable)
    __attribute__((aligned(16))) uint32_t marks[1];     // (the two mark bitmaps of the game under work follow: gc_marks_in_lds)
};
static_assert(64 * WPB == 256, "GcLds::hist has one bin per thread of a collector workgroup");
// k_sim_step's workgroups are five per CU (96 registers a wave), and gfx950 hands out its 160 KB of LDS in granules of 1 280
// bytes: 25 granules = 32 000 bytes a workgroup (at 32 032 bytes a CU held four - measured, scripts/launch_timeline.py: an
// eighth of the grid's workgroups started when the first ones had finished).  The marker keeps both bitmaps of the game it
// marks in LDS when they fit beside GcLds (N <= 100 000: 2 x 12.5 KB); beyond that they stay in memory (same code, atomics).
constexpr size_t GC_LDS_GRANULE = 1280, GC_LDS_MAX = (163840 / 5 / GC_LDS_GRANULE) * GC_LDS_GRANULE;
__host__ __device__ inline bool gc_marks_in_lds(int N) { return offsetof(GcLds, marks) + 2 * gc_bm_bytes(N) <= GC_LDS_MAX; }
__host__ __device__ inline size_t gc_lds_bytes(int N) { return offsetof(GcLds, marks) + (gc_marks_in_lds(N) ? 2 * gc_bm_bytes(N) : 16); }
static_assert(offsetof(GcLds, marks) + 2 * 12512 <= GC_LDS_MAX, "the reference's pool (ValueSim.py:16: 100 000 nodes) is marked in LDS");
constexpr int GC_MARK_WGS_MAX = 8;      // marking workgroups a game gets at most (each owns a share of its index range)

// thread 0 of a workgroup, after the workgroup's stores for this game: returns true for the last workgroup to arrive
__device__ __forceinline__ bool gc_arrive(int32_t* gs, int n_gc, bool leftover, bool& any_left, bool blocking) {
    const int add = 1 + (leftover ? 256 : 0);
    const int old = atomicAdd(&gs[TM_GS_GC_ARRIVE], add);
    if ((old & 255) != n_gc - 1) return false;
    any_left = ((old + add) >> 8) != 0;
    atomicExch(&gs[TM_GS_GC_ARRIVE], 0);
    if (blocking) gs[TM_GS_GC_SLICES] += 1;      // launches the game has waited for its collections
    return true;
}


// ---------------------------------------------------------------------------------------------------
// THE MARKER (core.h:32-50 get_all_childs, agents/agent.py:206-224 - the same set by another schedule; the schedule's core is
// restated and checked against the reference's set in oracle/uct_oracle.c orc_sweep_marks).
// The reference walks the tree breadth first; on the device that walk's duration is its DEPTH (40-80 levels of dependent round
// trips, r03-r05: 17-30 launches per marking).  Reachability is a fixpoint, and where a node's children are does not depend on
// what is marked: the child rows of a game are one contiguous array (32 B a node: seven children + the node's observation), and
// free indices are popped highest first, so children mostly lie BELOW their parents.  So the index range is cut into BLOCKS of
// 256 nodes, a game's marking workgroups each OWN a contiguous share of it (whole chunks of 1024 nodes or more: the unit the
// owners' flags in the control block speak of), and an owner keeps two bitmaps of its share in LDS - MARKED, and PENDING
// (marked, children not looked at yet) - and a work list of flagged blocks.  Its four WAVES work on their own, each on the
// highest flagged block below where it was last (descending sweeps), with no workgroup barrier between them - one wave's
// memory round trip is the others' work:
//   * the block's child rows are loaded - all of them, coalesced, whatever is pending: GC_ROWS rows a lane, in registers -
//     together with the block's words of the game's GLOBAL pending bitmap (taken by exchange: what other owners, earlier
//     launches and the game's write barrier have sent; a bit the owner has marked already is dropped);
//   * then the block is brought to ITS fixpoint: every pending node of the block is taken off the pending bitmap and marks
//     its children - one LDS atomic with return for a child of the owner's share (not marked before: pending now; of this
//     block: another round; of another block: that block into the work list, for whichever wave comes first) - and its
//     observation (a device atomic into the global bitmap, nothing comes back); again while a node OF THIS BLOCK became
//     pending.  Parents and children inside a block cost LDS round trips, not memory ones; a node's row is looked at once;
//   * a child in ANOTHER owner's share is SENT: its bit into the global pending bitmap, then - after s_waitcnt vmcnt(0), the
//     write-through-then-flag hand-off of MI355X_MICROARCH.md - its chunk's flag in the game's control block (TM_GS_GC_FLAGS).
//     Nothing comes back and nobody waits: an owner without flagged blocks polls its flags a few times, then leaves.
// The marking is complete when a launch ends with no flag up and nothing left by anybody.  It is resumable at block boundaries
// (the launch's deadline): marks go back to the global bitmap (a node that is still pending goes back pending, not marked),
// pending bits to the global pending bitmap (the head of the game's gc_queue words), flags to the control block; nothing else
// is kept, and the next launch may cut the shares differently.
// Beside a simulating game (speculative marking) the game's write barrier is one more sender, and ORs the marks of the nodes it
// creates into the global bitmaps while this runs: an owner merges its marks (atomicOr) instead of storing them.
// Node 0 is marked when a processed row holds a zero (the reference follows zero entries like any child: every leaf's row) and
// is never looked at itself (its own row is zero).
// LDSM = false: pools whose bitmaps do not fit the LDS (gc_marks_in_lds) - the same with all three bitmaps in memory (device-
// scope atomics, loads that bypass the first-level cache, one row a lane).  Correct, and not fast: no benchmarked configuration
// runs it.
// ---------------------------------------------------------------------------------------------------
constexpr int GC_IDLE_POLLS = 12;      // looks at its flags an owner without work takes before it leaves the launch
template <bool LDSM>
__device__ __forceinline__ bool gc_sweep_mark(const tm_store& S, const GP& P, GcLds& M, uint32_t* lds_marks, int part, int n_parts,
                                              bool spec, long long deadline, int tid, int* sm) {
    constexpr int T = 64 * WPB, R = LDSM ? GC_ROWS : 1, BLOCK_LOG2 = LDSM ? GC_BLOCK_LOG2 : 6, BLOCK = 1 << BLOCK_LOG2;
    constexpr int FW = TM_GC_FLAG_WORDS;
    static_assert(BLOCK == R * 64 && FW == 4, "a block is a wave's: R rows a lane");
    const int N = S.max_nodes, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int32_t* gs = P.gs();
    const size_t bm_bytes = gc_bm_bytes(N);
    const int nw = (int)(bm_bytes / 4);                  // words of one bitmap
    uint32_t* gnm = reinterpret_cast<uint32_t*>(S.gc_mark + (size_t)P.g * 2 * bm_bytes);
    uint32_t* gom = gnm + nw;
    uint32_t* gpd = reinterpret_cast<uint32_t*>(S.gc_queue + (size_t)P.g * N);
    uint32_t* nm = LDSM ? lds_marks : gnm;
    uint32_t* pd = LDSM ? lds_marks + nw : gpd;
    // chunks (the control block's flags), units (the work list's; a block, or - very large pools - a power of two of blocks)
    const int clog = gc_chunk_log2(N);
    int ulog = BLOCK_LOG2;
    while (((N + (1 << ulog) - 1) >> ulog) > 32 * GC_WL_WORDS) ++ulog;
    const int upc_log = clog - ulog, bpu_log = ulog - BLOCK_LOG2;
    const int n_chunks = (N + (1 << clog) - 1) >> clog, n_units = (N + (1 << ulog) - 1) >> ulog;
    // this owner's share: chunks [c_lo, c_hi), nodes [n_lo, n_hi), bitmap words [w_lo, w_hi)
    const int c_lo = (int)((long long)n_chunks * part / n_parts), c_hi = (int)((long long)n_chunks * (part + 1) / n_parts);
    const int n_lo = c_lo << clog, n_hi = min(N, c_hi << clog);
    const int w_lo = n_lo >> 5, w_hi = (n_hi + 31) >> 5;
    auto rd = [&](const uint32_t* p, uint32_t w) -> uint32_t {          // (a bitmap word that other waves are setting bits in)
        return __hip_atomic_load(p + w, __ATOMIC_RELAXED, LDSM ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT);
    };
    auto lds_rd = [&](const void* p) -> int { return __hip_atomic_load(reinterpret_cast<const int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    // One word of the game's global pending bitmap (what other owners, earlier launches and the game's write barrier have sent to
    // 32 nodes of this share), taken by exchange: a node this owner has marked already has been, or will be, looked at - the
    // others are marked and pending now, and their block goes into the work list.
    auto take_word = [&](int iw) {
        if (LDSM) {
            const uint32_t x = atomicExch(gpd + iw, 0u);
            if (x) {
                const uint32_t fresh = x & ~rd(nm, (uint32_t)iw);
                atomicOr(nm + iw, x);
                if (fresh) { atomicOr(pd + iw, fresh); const int u = (iw << 5) >> ulog; atomicOr(&M.wl[u >> 5], 1u << (u & 31)); }
            }
        } else if (rd(pd, (uint32_t)iw)) {      // (in memory the pending bitmap is the global one: the bits are where they belong)
            const int u = (iw << 5) >> ulog;
            atomicOr(&M.wl[u >> 5], 1u << (u & 31));
        }
    };
    // the flags of this share that are up in the control block come down there and wait in LDS (GcLds::dirty) for a wave to
    // take the flagged chunk's words of the global pending bitmap (lanes 0..3 of one wave)
    auto poll_flags = [&]() {
        if (lane < FW) {
            const uint32_t mk = M.rmask[lane];
            const uint32_t got = mk ? (uint32_t)atomicAnd(&gs[TM_GS_GC_FLAGS + lane], (int)~mk) & mk : 0u;
            if (got) atomicOr(&M.dirty[lane], got);
        }
    };
    // ---- the marks of this share so far, and its blocks that hold pending nodes ----
    {
        if (LDSM) for (int i = w_lo + tid; i < w_hi; i += T) { nm[i] = gnm[i]; pd[i] = 0u; }
        if (tid < FW) {
            uint32_t mk = 0;
            for (int b = 0; b < 32; ++b) { const int ck = tid * 32 + b; if (ck >= c_lo && ck < c_hi) mk |= 1u << b; }
            M.rmask[tid] = mk;
            M.dirty[tid] = mk ? (uint32_t)atomicAnd(&gs[TM_GS_GC_FLAGS + tid], (int)~mk) & mk : 0u;      // the share's flags that are up
        }
        if (tid < WPB * FW) M.fsend[tid / FW][tid % FW] = 0u;
        if (tid < GC_WL_WORDS) M.wl[tid] = 0u;
        if (tid < 3) {                 // busy, quit, zero (a zero made here: hoisted out of the games' loop it was the kernel's only spill)
            int z; asm volatile("v_mov_b32 %0, 0" : "=v"(z));
            (&M.busy)[tid] = z;
        }
        if (tid == 0) gs[TM_GS_GC_MARK_LAUNCHES] += part == 0 ? 1 : 0;
        __syncthreads();
    }
    const uint4* kids4 = reinterpret_cast<const uint4*>(P.kids());
    int cur = 32 * GC_WL_WORDS;        // this wave's position: units below it are still to come in its sweep
    int n_blocks = 0, n_iters = 0, idle = 0, n_idle = 0;
    long long cyc_load = 0, cyc_rounds = 0;
    const long long t_begin = (long long)__builtin_readcyclecounter();
    // ---- the waves, each on its own ----
    for (;;) {
        if (deadline >= 0 && (long long)__builtin_readcyclecounter() > deadline) break;
        // (the count of waves at work is read BEFORE the lists: a wave that has left the count has flagged what it found)
        const int busy0 = lds_rd(&M.busy);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // a chunk whose flag has come down from the control block: its words of the global pending bitmap, one a lane
        {
            const uint32_t df = lane < FW ? (uint32_t)lds_rd(&M.dirty[lane < FW ? lane : 0]) : 0u;
            const uint64_t dm = __ballot(df != 0u);
            if (dm) {
                const int di = __ffsll((long long)dm) - 1;
                const uint32_t dw = rl_u32(df, di);
                const int ck = 32 * di + 31 - __clz((int)dw);
                uint32_t oldd = 0;
                if (lane == 0) { atomicAdd(&M.busy, 1); oldd = atomicAnd(&M.dirty[di], ~(1u << (ck & 31))); }
                oldd = rl_u32(oldd, 0);
                if ((oldd >> (ck & 31)) & 1u) {
                    for (int w0 = lane; w0 < (1 << (clog - 5)); w0 += 64) { const int iw = (ck << (clog - 5)) + w0; if (iw < nw) take_word(iw); }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                }
                if (lane == 0) atomicSub(&M.busy, 1);
                continue;
            }
        }
        const uint32_t v = lane < GC_WL_WORDS ? (uint32_t)lds_rd(&M.wl[lane < GC_WL_WORDS ? lane : 0]) : 0u;
        const int lim = cur - 32 * lane;                                 // bits of word `lane` below the position
        const uint32_t vb = lim >= 32 ? v : lim <= 0 ? 0u : (v & ((1u << lim) - 1u));
        uint64_t mb = __ballot(vb != 0u);
        uint32_t word;
        int wi;
        if (mb) { wi = 63 - __clzll((long long)mb); word = rl_u32(vb, wi); }
        else {
            mb = __ballot(v != 0u);                                       // the next sweep: from the highest flagged unit down
            wi = mb ? 63 - __clzll((long long)mb) : 0;
            word = mb ? rl_u32(v, wi) : 0u;
        }
        if (word == 0u) {
            n_idle += 1;
            // nothing flagged in this share.  Another wave of the workgroup may still flag something ...
            if (busy0 != 0) { __builtin_amdgcn_s_sleep(4); continue; }
            // ... or another owner, or the game's barrier (what the barrier flags is the next launch's when it is the only sender)
            if (n_parts == 1) break;
            if (lds_rd(&M.quit)) break;
            if (wave == 0) {
                if (++idle > GC_IDLE_POLLS) { if (lane == 0) atomicExch(&M.quit, 1); break; }
                __builtin_amdgcn_s_sleep(32);
                poll_flags();
            } else __builtin_amdgcn_s_sleep(16);
            continue;
        }
        const int u = 32 * wi + 31 - __clz((int)word);
        // claim it (another wave may be faster)
        uint32_t oldw = 0;
        if (lane == 0) { atomicAdd(&M.busy, 1); oldw = atomicAnd(&M.wl[u >> 5], ~(1u << (u & 31))); }
        oldw = rl_u32(oldw, 0);
        if (!((oldw >> (u & 31)) & 1u)) { if (lane == 0) atomicSub(&M.busy, 1); continue; }
        cur = u;
        idle = 0;
        for (int b = (1 << bpu_log) - 1; b >= 0; --b) {
            const int blk = (u << bpu_log) + b, base = blk << BLOCK_LOG2;
            if (base >= N) continue;
            n_blocks += 1;
            const long long tb0 = (long long)__builtin_readcyclecounter();
            uint4 k0[R], k1[R];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int node = base + j * 64 + lane;
                const bool ok = node < N;
                k0[j] = ok ? kids4[(size_t)node * 2] : make_uint4(0, 0, 0, 0);
                k1[j] = ok ? kids4[(size_t)node * 2 + 1] : make_uint4(0, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long tb1 = (long long)__builtin_readcyclecounter();
            cyc_load += tb1 - tb0;
            for (;;) {
                uint32_t again = 0, zero_seen = 0;
                n_iters += 1;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const int node = base + j * 64 + lane;         // (the wave's lanes: 64 consecutive nodes = two words)
                    const uint32_t pw = rd(pd, (uint32_t)(node < N ? node : 0) >> 5);
                    const uint32_t pend = (node < N ? 1u : 0u) & (pw >> (node & 31)) & 1u;
                    const uint64_t pm = __ballot((int)pend);
                    if (pm == 0ull) continue;                      // (after the first round few slots have anything)
                    if ((lane & 31) == 0) {
                        const uint32_t half = lane ? (uint32_t)(pm >> 32) : (uint32_t)pm;
                        if (half) atomicAnd(pd + ((uint32_t)node >> 5), ~half);
                        if (!LDSM && half) atomicOr(nm + ((uint32_t)node >> 5), half);      // (in LDS: marked when it was taken)
                    }
                    // Most rows are a leaf's (all zero), and of a block's nodes few are pending at a time: the children are not marked
                    // seven a lane, but EIGHT PENDING EXPANDED NODES AT A TIME, one child a lane (lane l: child l & 7 of the group's
                    // node l >> 3; 56 of the 64 lanes) - the instructions a block costs are what bounds the marker (a wave's
                    // round trip for the rows: 1 300 cycles; the marking of a block with seven children a lane: 12 000).
                    const bool expd = pend && (k0[j].x | k0[j].y | k0[j].z | k0[j].w | k1[j].x | k1[j].y | k1[j].z) != 0u;
                    zero_seen |= (pend && !expd) ? 1u : 0u;                  // (a leaf: its row's zeros reach node 0)
                    const uint64_t em = __ballot(expd);
                    if (em != 0ull) {
                        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
                        const int n_exp = __popcll(em);
                        // lane r gets the lane of the node of rank r (the lanes that have nothing to push push to lane 63, which is
                        // read only when all 64 have): one push for the slot, a pull a group
                        const int ranked = __builtin_amdgcn_ds_permute((expd ? rank : 63) << 2, lane);
                        const int a = lane & 7;
                        // one child a lane out of the rows of the group's eight nodes
                        auto gather = [&](int g) -> uint32_t {
                            const int src = __builtin_amdgcn_ds_bpermute((8 * g + (lane >> 3)) << 2, ranked) << 2;
                            uint32_t cn = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)k0[j].x);
                            { const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)k0[j].y); cn = a == 1 ? t : cn; }
                            { const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)k0[j].z); cn = a == 2 ? t : cn; }
                            { const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)k0[j].w); cn = a == 3 ? t : cn; }
                            { const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)k1[j].x); cn = a == 4 ? t : cn; }
                            { const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)k1[j].y); cn = a == 5 ? t : cn; }
                            { const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)k1[j].z); cn = a == 6 ? t : cn; }
                            const bool valid = 8 * g + (lane >> 3) < n_exp && a < 7;
                            zero_seen |= (valid && cn == 0u) ? 1u : 0u;
                            return valid ? cn : 0u;
                        };
                        auto is_own = [&](uint32_t cn) { return cn != 0u && (LDSM ? ((int)cn >= n_lo && (int)cn < n_hi) : true); };
                        // what a test-and-set found
                        auto settle = [&](uint32_t cn, uint32_t old) {
                            if (cn == 0u) return;
                            const uint32_t bit = 1u << (cn & 31);
                            const int cb = (int)(cn >> BLOCK_LOG2), cu = cb >> bpu_log, cc = (int)(cn >> clog);
                            const bool own = is_own(cn);
                            bool flag_other = !own;
                            if (own && !(old & bit)) {
                                atomicOr(pd + (cn >> 5), bit);
                                // this block's own next round meets a node of the block; a lower block of the unit under work is
                                // still to come; every other block of this share: into the work list
                                if (cb == blk) again = 1;
                                else if (!(cu == u && cb < blk)) {
                                    if (LDSM || ((M.rmask[cc >> 5] >> (cc & 31)) & 1u)) atomicOr(&M.wl[cu >> 5], 1u << (cu & 31));
                                    else flag_other = true;
                                }
                            } else if (!own) atomicOr(gpd + (cn >> 5), bit);        // sent: the owner of that share takes it from there
                            // another owner's chunk: its flag goes up in the control block once the pending bit is in memory
                            if (flag_other) atomicOr(&M.fsend[wave][cc >> 5], 1u << (cc & 31));
                        };
                        // two groups a turn: their LDS round trips (pull, gather, test-and-set: a group is a chain of them) overlap
                        for (int g = 0; 8 * g < n_exp; g += 2) {
                            const uint32_t ca = gather(g), cb2 = gather(g + 1);
                            uint32_t oa = 0xFFFFFFFFu, ob = 0xFFFFFFFFu;
                            if (is_own(ca)) oa = atomicOr(nm + (ca >> 5), 1u << (ca & 31));        // (only the lanes that have one: LDS atomics of
                            if (is_own(cb2)) ob = atomicOr(nm + (cb2 >> 5), 1u << (cb2 & 31));    //  many lanes on one address are serialised)
                            settle(ca, oa);
                            settle(cb2, ob);
                        }
                    }
                    const uint32_t o = pend ? k1[j].w : 0u;                  // the node's observation (0 in a slot that is being created)
                    if (o != 0u) atomicOr(gom + (o >> 5), 1u << (o & 31));
                }
                if (__any((int)zero_seen) && lane == 0) atomicExch(&M.zero, 1);
                if (!LDSM) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (!__any((int)again)) break;
            }
            cyc_rounds += (long long)__builtin_readcyclecounter() - tb1;
        }
        // the flags of other owners' chunks: up in the control block once every pending bit this wave sent is in memory
        {
            const uint32_t f = lane < FW ? (uint32_t)lds_rd(&M.fsend[wave][lane < FW ? lane : 0]) : 0u;
            if (__any(f != 0u)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (f) { atomicAnd(&M.fsend[wave][lane], ~f); atomicOr(&gs[TM_GS_GC_FLAGS + lane], (int)f); }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // (what this wave flagged is in the work list before it leaves the count)
        if (lane == 0) atomicSub(&M.busy, 1);
    }
    // ---- what is left, and the marks back where the other steps (and the next launch) read them ----
    __syncthreads();
    if (tid == 0 && M.zero) { atomicOr(gnm, 1u); atomicOr(gom, 1u); }     // (word 0 of the global bitmap, whoever owns it)
    if (tid < GC_WL_WORDS) {           // the work list's units back into chunk flags
        for (uint32_t left = M.wl[tid]; left; left &= left - 1) {
            const int ck = (32 * tid + __ffs((int)left) - 1) >> upc_log;
            atomicOr(&M.dirty[ck >> 5], 1u << (ck & 31));
        }
    }
    __syncthreads();
    bool leftover = false;
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < FW; ++i) {
            const uint32_t l = M.dirty[i];
            if (l) { leftover = true; atomicOr(&gs[TM_GS_GC_FLAGS + i], (int)l); }
        }
        atomicAdd(&gs[TM_GS_GC_MARK_CYC], (int)(((long long)__builtin_readcyclecounter() - t_begin) >> 6));
    }
    if (lane == 0) {
        atomicAdd(&gs[TM_GS_GC_BLOCKS], n_blocks); atomicAdd(&gs[TM_GS_GC_ITERS], n_iters); atomicAdd(&gs[TM_GS_GC_IDLE_TURNS], n_idle);
        atomicAdd(&gs[TM_GS_GC_CYC_LOAD], (int)(cyc_load >> 6)); atomicAdd(&gs[TM_GS_GC_CYC_ROUNDS], (int)(cyc_rounds >> 6));
        atomicAdd(&gs[TM_GS_GC_CYC_WAVES], (int)(((long long)__builtin_readcyclecounter() - t_begin) >> 6));
    }
    if (LDSM) {
        for (int i = w_lo + tid; i < w_hi; i += T) {
            // A node that is still pending (the deadline came before its block) goes back PENDING, NOT MARKED: the next
            // launch's owner drops what it is sent and finds marked ("looked at, or pending with me" - its own LDS is gone by then).
            const uint32_t q = pd[i], l = nm[i] & ~q;
            // (bit 0 of word 0 is node 0's: set above by whoever saw a zero; the game's write barrier is setting bits too)
            if (!spec && i != 0) gnm[i] = l;
            else if (l & ~gnm[i]) atomicOr(gnm + i, l);
            if (q) atomicOr(gpd + i, q);                     // (nodes of the blocks that are still flagged)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (what this workgroup sent and flagged is in memory before it arrives)
    __syncthreads();                   // (the LDS bitmaps may be the next game's now)
    return leftover;                   // (thread 0's)
}

__device__ __forceinline__ void gc_collector_block(const tm_store& S, int flags, int n_gc, GcLds& M) {
    constexpr int T = 64 * WPB;
    typedef Grp<T> G_;
    const int tid = threadIdx.x, lane = tid & 63, c = blockIdx.x;
    const int seq = (int)((unsigned)flags >> 8);
    const long long budget = ((flags & TM_SIM_GC_FULL) || S.gc_slice_cycles <= 0) ? -1 : (long long)S.gc_slice_cycles;
    const long long deadline = budget < 0 ? -1 : (long long)__builtin_readcyclecounter() + budget;
    int* sm = M.scan;
    // ---- which games are collecting (requested in an earlier launch, or under way), in game order: first the games that
    // are waiting for their collection (at most GC_LIST_WAIT), then the speculative markings ----
    // thread t looks at a contiguous run of games (all its loads in flight together, one prefix sum for the whole list)
    int n_wait = 0, n_spec = 0;
    constexpr int RUN = 16;
    // (a phase word that the game's wave replaces in the middle of a launch reads as what it replaced, see GC_REQ_OVER)
    auto classify = [&](int& word) -> int {       // 0: waiting for its collection, 1: speculative marking, -1: neither
        const int ph = word & 15;
        const bool now = (word >> 4) == seq;
        if (ph >= GCP_MARK && ph <= GCP_LAST) return 0;
        if (ph == GC_REQ) return now ? -1 : 0;
        if (ph == GC_SPEC_REQ) return now ? -1 : 1;
        if (ph == GC_SPEC_MARK) return 1;
        if (ph == GC_REQ_SPEC) { if (now) { word = GC_SPEC_MARK; return 1; } return 0; }
        if (ph == GC_REQ_OVER) {
            // (the compare-and-swap to GC_SPEC_MARK at the end of the step fails whichever word the last arriver expects)
            if (now) { word = (word & ~15) | GC_SPEC_REQ; return 1; }
            return 0;
        }
        return -1;
    };
    // a thread's run of RUN games: the summary words of its four groups of four (TM_GS_GC_ACTIVE4), then the phase words of the
    // games they point at (in flight together); everybody else's reads as zero
    static_assert(RUN == 16, "a run = four groups of four games");
    auto load_run = [&](int base, int (&word)[RUN]) {
        const int g0 = base + tid * RUN;
        uint32_t act = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (g0 + 4 * q < S.n_games) act |= ((uint32_t)S.gs[(size_t)(g0 + 4 * q) * TM_GS_DW + TM_GS_GC_ACTIVE4] & 15u) << (4 * q);
#pragma unroll
        for (int r = 0; r < RUN; ++r)
            word[r] = ((act >> r) & 1u) && g0 + r < S.n_games ? S.gs[(size_t)(g0 + r) * TM_GS_DW + TM_GS_GC_PHASE] : 0;
    };
    int n_wait_all = 0;
    for (int base = 0; base < S.n_games; base += T * RUN) {
        int word[RUN];
        load_run(base, word);
        uint32_t todo_w = 0, todo_s = 0;
#pragma unroll
        for (int r = 0; r < RUN; ++r) {
            const int cls = classify(word[r]);
            if (cls == 0) todo_w |= 1u << r;
            if (cls == 1) todo_s |= 1u << r;
        }
        int total_w, total_s;
        int pos_w = n_wait_all + G_::exscan(__popc(todo_w), tid, sm, total_w);
        int pos_s = GC_LIST_WAIT + n_spec + G_::exscan(__popc(todo_s), tid, sm, total_s);
        for (uint32_t bits = todo_w; bits; bits &= bits - 1) {
            const int r = __ffs(bits) - 1;
            if (pos_w < GC_LIST_WAIT) { M.list_g[pos_w] = base + tid * RUN + r; M.list_ph[pos_w] = word[r]; }
            pos_w += 1;
        }
        for (uint32_t bits = todo_s; bits; bits &= bits - 1) {
            const int r = __ffs(bits) - 1;
            if (pos_s < GC_LIST_MAX) { M.list_g[pos_s] = base + tid * RUN + r; M.list_ph[pos_s] = word[r]; }
            pos_s += 1;
        }
        n_wait_all += total_w;
        n_spec = min(GC_LIST_MAX - GC_LIST_WAIT, n_spec + total_s);
    }
    n_wait = min(GC_LIST_WAIT, n_wait_all);
    if (n_wait_all > GC_LIST_WAIT) {
        // More games wait than a launch looks after: the ones that have waited LONGEST get the places (in game-index order the
        // high-numbered games of a big batch waited for all the others, and a move's catch-up is as long as its slowest game).
        // A histogram of the waiting times (launches since the request, capped) gives the cut-off; the games beyond it, and of
        // those at it the first in index order, are taken.  Every workgroup reads the same words: the same list everywhere.
        __syncthreads();
        M.hist[tid] = 0;
        __syncthreads();
        auto age_of = [&](int g) { return min(255, (seq - S.gs[(size_t)g * TM_GS_DW + TM_GS_GC_REQ_AT]) & 0x7FFFF); };
        for (int base = 0; base < S.n_games; base += T * RUN) {
            int word[RUN];
            load_run(base, word);
#pragma unroll
            for (int r = 0; r < RUN; ++r) {
                const int g = base + tid * RUN + r;
                if (classify(word[r]) == 0) atomicAdd(&M.hist[age_of(g)], 1);
            }
        }
        __syncthreads();
        int cut = 255, older = 0;              // older = games that have waited longer than `cut` launches (< GC_LIST_WAIT of them)
        for (; cut > 0 && older + M.hist[cut] < GC_LIST_WAIT; --cut) older += M.hist[cut];
        __syncthreads();
        int n_old = 0, n_eq = 0;
        for (int base = 0; base < S.n_games; base += T * RUN) {
            int word[RUN];
            load_run(base, word);
            uint32_t m_old = 0, m_eq = 0;
#pragma unroll
            for (int r = 0; r < RUN; ++r) {
                const int g = base + tid * RUN + r;
                if (classify(word[r]) == 0) {
                    const int a = age_of(g);
                    if (a > cut) m_old |= 1u << r;
                    else if (a == cut) m_eq |= 1u << r;
                }
            }
            int t_old, t_eq;
            int p_old = n_old + G_::exscan(__popc(m_old), tid, sm, t_old);
            int p_eq = older + n_eq + G_::exscan(__popc(m_eq), tid, sm, t_eq);
            for (uint32_t bits = m_old; bits; bits &= bits - 1) {
                const int r = __ffs(bits) - 1;
                M.list_g[p_old] = base + tid * RUN + r; M.list_ph[p_old] = word[r];
                p_old += 1;
            }
            for (uint32_t bits = m_eq; bits; bits &= bits - 1) {
                const int r = __ffs(bits) - 1;
                if (p_eq < GC_LIST_WAIT) { M.list_g[p_eq] = base + tid * RUN + r; M.list_ph[p_eq] = word[r]; }
                p_eq += 1;
            }
            n_old += t_old;
            n_eq += t_eq;
        }
    }
    __syncthreads();
    const int n_list = n_wait + n_spec;
    if (n_list == 0) return;
    {   // the speculative markings right behind the waiting games
        int mg = 0, mp = 0;
        if (tid < n_spec) { mg = M.list_g[GC_LIST_WAIT + tid]; mp = M.list_ph[GC_LIST_WAIT + tid]; }
        __syncthreads();
        if (tid < n_spec) { M.list_g[n_wait + tid] = mg; M.list_ph[n_wait + tid] = mp; }
        __syncthreads();
    }
    const int N = S.max_nodes;
    const size_t bm_bytes = gc_bm_bytes(N);
    const uint32_t mask = (uint32_t)S.table_cap - 1u;
    const int n_words = (N + 31) / 32;
    const bool dharvest = S.online && S.replay_cap > 0 && S.kind == TM_KIND_DIST && S.replay_dist;
    const bool harvest = S.online && S.replay_cap > 0 && S.kind != TM_KIND_DIST;
    // THE LAUNCH'S PLAN (every workgroup derives the same one: the control words it reads were written in earlier launches,
    // or are written in this one only after every workgroup has arrived for the game).
    // The bounded steps first, as many as fit the launch's cost allowance - those of the waiting games oldest request first,
    // then the speculative markings' starting at a game that rotates with the launch number; then the markings: the marking
    // workgroups are dealt to the marked games in the same order - up to GC_MARK_WGS_MAX a game, each the owner of a share of
    // the game's index range (gc_sweep_mark: the bitmaps of its share live in its LDS); more games than workgroups: one
    // workgroup a game, its time shared between its games.
    // Workgroups [0, n_b) do the bounded steps, [n_b, n_gc) the marking (a lone workgroup does both): a launch's bounded work
    // must not take the marking's time, nor the other way round.  The step that follows a speculative marking (GC_REQ_SPEC:
    // the game has stopped, what its barrier sent is all there) is both: the tables are cleared by the bounded workgroups while
    // the game's marking workgroups take the marking to its end.
    // EVERY workgroup arrives for every game whose step is performed, with or without a share of the work: nobody moves a
    // game on before everybody has read its control words.  Arrivals without a share are made first, one thread per game,
    // all in flight together (a speculative marking with nothing flagged - GC_IDLE - is nothing but that: the last arriver
    // looks at the flags the game's barrier has raised meanwhile).
    const int n_b = n_gc >= 2 ? n_gc / 2 : 1, n_m = n_gc >= 2 ? n_gc - n_b : 1;
    if (tid < n_list) {
        const int ph = M.list_ph[tid] & 15;
        const int32_t* gsk = S.gs + (size_t)M.list_g[tid] * TM_GS_DW;
        int step = ph == GC_REQ_OVER ? GCP_INIT : ph;   // the step this launch performs for the game
        if (ph == GC_SPEC_MARK) step = gsk[TM_GS_GC_WORK] != 0 ? GCP_MARK : GC_IDLE;
        // a collection under way was begun by launches with n_gc collector workgroups - its shares, its counts per workgroup and
        // its arrival count are cut for that many.  Met by a launch with another number (another range of games: a sub-batch
        // here, the whole store there) it is left alone and the game is flagged: a misuse must raise, not corrupt free lists.
        if (step != GCP_INIT && step != GC_SPEC_REQ && gsk[TM_GS_GC_NGC] != n_gc) {
            step = GC_FOREIGN;
            if (c == 0) atomicOr(&S.gs[(size_t)M.list_g[tid] * TM_GS_DW + TM_GS_ERR], TM_ERR_GC_GRID);
        }
        M.list_step[tid] = step;
        M.list_mark[tid] = 0;
        if (tid < n_wait) M.age[tid] = (seq - gsk[TM_GS_GC_REQ_AT]) & 0x7FFFF;      // launches since the game asked (the launch number has 19 bits)
    }
    __syncthreads();
    if (tid < n_wait) {        // the waiting games, oldest request first
        const int mine = M.age[tid];
        int rank = 0;
        for (int j = 0; j < n_wait; ++j) { const int other = M.age[j]; rank += (other > mine || (other == mine && j < tid)) ? 1 : 0; }
        M.by_age[rank] = (short)tid;
    }
    __syncthreads();
    // The plan, one list entry a thread (position p of the plan's order = the waiting games by age, then the speculative markings
    // from a game that rotates with the launch number), a few prefix sums over the workgroup, and ONE short loop of one thread:
    // the greedy choice of the bounded steps, over the entries that have one.  (Until r06 the whole plan was one thread's loop
    // over the list - 128 000 cycles of dependent LDS accesses with a full list, the launch's whole allowance: with two hundred
    // games waiting no marker had time left, and the trained net's steady state stayed there.)
    const int cm = n_gc >= 2 ? c - n_b : 0;           // index among the marking workgroups (< 0: not one of them)
    {
        const int p = tid;
        const bool in_plan = p < n_list;
        const int k = !in_plan ? 0 : p < n_wait ? (int)M.by_age[p] : n_wait + ((p - n_wait) + seq) % n_spec;
        const int step = in_plan ? M.list_step[k] : GC_FOREIGN;
        const bool bounded = in_plan && step != GC_FOREIGN && step != GC_IDLE && step != GCP_MARK;
        const int cost = !bounded ? 0 : step == GCP_WRITE ? 7 : step == GCP_COUNT ? 6 : 1;      // init / clear: 1; count + nodes: 1 + 5; write + observations: 2 + 5
        // 1. the entries with a bounded step, in the plan's order, and one thread's greedy pass over them
        int n_bd;
        const int bpos = G_::exscan(bounded ? 1 : 0, tid, sm, n_bd);
        if (bounded) { M.order[bpos] = (short)k; M.mk_list[bpos] = (short)cost; }        // (both lists are rebuilt below)
        if (in_plan) M.list_parts[k] = bounded ? 0 : 1;      // (a bounded step: until the greedy pass has taken it)
        __syncthreads();
        if (tid == 0) {
            int cost_left = deadline < 0 ? 1 << 20 : (S.gc_cost_units > 0 ? S.gc_cost_units : GC_COST_MAX);      // (collector-only launches: nothing to hold up)
            for (int i = 0; i < n_bd && cost_left > 0; ++i) {
                const int ci = M.mk_list[i];
                if (ci <= cost_left) { cost_left -= ci; M.list_parts[M.order[i]] = (short)n_b; }      // performed (its shares: the bounded workgroups)
            }
        }
        __syncthreads();
        // 2. what is performed in this launch, and this workgroup's share of it
        const bool performed = in_plan && step != GC_FOREIGN && (!bounded || M.list_parts[k] != 0);
        __syncthreads();
        if (in_plan) {
            M.list_parts[k] = (short)(!performed ? 0 : bounded ? n_b : 1);      // 0: not in this launch, nobody arrives
            M.list_part[k] = (short)((performed && bounded && c < n_b) ? c : -1);
            M.list_share[k] = 0;
        }
        // 3. the marked games in the plan's order (the waiting ones first), and the marking workgroups round them
        const bool marked = performed && (step == GCP_MARK || step == GC_REQ_SPEC);
        int n_mk, n_mw;
        const int j = G_::exscan(marked ? 1 : 0, tid, sm, n_mk);
        G_::exscan((marked && p < n_wait) ? 1 : 0, tid, sm, n_mw);
        // A game that WAITS for its marking gets whole workgroups, up to GC_MARK_WGS_MAX, each the owner of a share of its index
        // range; the speculative markings get what is left the same way, or - more of them than workgroups - take turns on them:
        // a workgroup does its games one after the other, each until its work list is empty or the launch's deadline (most of them
        // have a few blocks to look at: what the game's barrier has sent since the last launch); the one it does not get to stays
        // as it is.  (r06, measured: with the workgroups dealt evenly and the time of a shared workgroup divided, every marking
        // ran on a fraction of a slice minus its fixed costs.)
        const int n_ms = n_mk - n_mw;
        // (a quarter of the workgroups stays with the speculative markings while there are any: a marking that is done when its
        // game's pool runs dry costs the game one launch, not the marking's)
        const int keep = min(n_ms, n_m / 4);
        const int per_w = n_mw == 0 ? 0 : min(GC_MARK_WGS_MAX, (n_m - keep) / n_mw);      // 0: more waiting games than workgroups
        const int used_w = n_mw == 0 ? 0 : per_w >= 1 ? n_mw * per_w : n_m - keep;
        const int rem = n_m - used_w;
        const int per_s = n_ms == 0 ? 0 : min(GC_MARK_WGS_MAX, rem / n_ms);              // 0: the speculative markings take turns (or wait)
        bool mine = false;
        if (marked) {
            const bool waiting = j < n_mw;
            const int jj = waiting ? j : j - n_mw, per = waiting ? per_w : per_s, first = waiting ? 0 : used_w, pool = waiting ? used_w : rem;
            if (per >= 1) {          // its own workgroups: [first + jj * per, first + (jj + 1) * per)
                if (cm >= first + jj * per && cm < first + (jj + 1) * per) {
                    mine = true; M.list_mpart[k] = (unsigned char)(cm - first - jj * per); M.list_mparts[k] = (unsigned char)per;
                }
                M.list_share[k] = 1;
            } else if (pool >= 1) {  // a workgroup of the pool, with others
                if (cm >= 0 && first + jj % pool == cm) { mine = true; M.list_mpart[k] = 0; M.list_mparts[k] = 1; }
                M.list_share[k] = (short)((waiting ? n_mw : n_ms) / pool + 1);
            } else if (step == GCP_MARK) {
                M.list_parts[k] = 0;             // no workgroup left for it in this launch: nobody arrives
            }
            // (GC_REQ_SPEC without a marking workgroup: the tables are cleared, the marking goes on in the next launch - its flags
            // are up)
            if (mine) M.list_mark[k] = 1;
        }
        __syncthreads();             // (the two lists of step 1 have been read)
        // 4. this workgroup's work, in the plan's order: its bounded shares, then its markings
        int n_sh, n_mine;
        const bool share = in_plan && M.list_parts[k] != 0 && M.list_part[k] >= 0;
        const int spos = G_::exscan(share ? 1 : 0, tid, sm, n_sh);
        const int mpos = G_::exscan((mine && !share) ? 1 : 0, tid, sm, n_mine);
        if (share) M.order[spos] = (short)k;
        if (mine && !share) M.order[n_sh + mpos] = (short)k;
        if (tid == 0) M.n_order = n_sh + n_mine;
    }
    __syncthreads();
    // a single thread, after its workgroup's stores for the game: arrive; the last workgroup to arrive moves the game on
    // (two uniform values the steps' ends store: taken from their scalar registers where they are stored - kept in vector
    // registers across the marking loop they were the kernel's only spills)
    auto vreg = [](int x) { int v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x)); return v; };
    auto arrive = [&](int k, bool leftover) {
        const int g = M.list_g[k], word_seen = M.list_ph[k], step = M.list_step[k];
        const bool spec = (word_seen & 15) == GC_SPEC_MARK;       // the game is simulating
        int32_t* gs = S.gs + (size_t)g * TM_GS_DW;
        bool any_left = false;
        if (!gc_arrive(gs, n_gc, leftover, any_left, !spec && step != GC_SPEC_REQ)) return;
        // what is left of a marking: what a marking workgroup said when it arrived, and the flags that are up now (raised for
        // a share whose owner had left already, or by the game's barrier; every marking workgroup's flags were in memory before
        // it arrived)
        auto flags_up = [&]() {
            int f = 0;
            for (int i = 0; i < TM_GC_FLAG_WORDS; ++i) f |= atomicOr(&gs[TM_GS_GC_FLAGS + i], 0);
            return f != 0;
        };
        if (step == GCP_INIT) {
            gs[TM_GS_GC_NGC] = vreg(n_gc);         // the collection is the work of launches with this many collector workgroups
            gs[TM_GS_GC_WORK] = 1;
            gs[TM_GS_GC_PHASE] = GCP_MARK;
        } else if (step == GC_SPEC_REQ) {
            gs[TM_GS_GC_NGC] = vreg(n_gc);
            gs[TM_GS_GC_WORK] = 1;
            // (the game's wave may have turned its request into a blocking one in this very launch - then that stands, and the
            // next launch initialises from scratch)
            atomicCAS(&gs[TM_GS_GC_PHASE], word_seen, GC_SPEC_MARK);
        } else if (step == GC_REQ_SPEC) {
            // the game has stopped: nothing is sent by its barrier any more, and its markers have just looked at what was
            gs[TM_GS_GC_PHASE] = (any_left || flags_up()) ? GCP_MARK : GCP_COUNT;
        } else if (step == GCP_MARK || step == GC_IDLE) {
            // a speculative marking never ends by itself (the game may flag chunks after this point), and it does not touch the
            // phase word (the game's wave may have put its blocking request there in this launch)
            const bool left = any_left || flags_up();
            if (spec) gs[TM_GS_GC_WORK] = left ? 1 : 0;
            else if (!left) gs[TM_GS_GC_PHASE] = GCP_COUNT;
        } else if (step == GCP_COUNT) {
            gs[TM_GS_GC_PHASE] = GCP_WRITE;
        } else if (step == GCP_WRITE) {
            // the totals, from the counts of the launch before (nothing is exchanged inside a launch)
            const int32_t* part = S.gc_part + (size_t)g * TM_GC_PART_DW;
            int nf = 0, of = 0, kf = 0;
            for (int b = 0; b < n_b; ++b) { nf += part[3 * b]; of += part[3 * b + 1]; kf += part[3 * b + 2]; }
            gs[TM_GS_NFREE_NODE] = nf;
            gs[TM_GS_NFREE_OBS] = of;
            gs[TM_GS_CYC_TAIL + 1] = N - nf;      // reachable nodes at the last GC
            if (harvest || dharvest) {
                const int m0 = S.replay_count[g];
                if (m0 + kf > S.replay_cap) gs[TM_GS_N_DROPPED] += m0 + kf - S.replay_cap;   // never silently (the reference keeps all up to memory_size)
                S.replay_count[g] = min(S.replay_cap, m0 + kf);
            }
            // (the kept observations went back into their table in this launch, the kept nodes in the one before: complete)
            gs[TM_GS_N_GC] += 1;
            gs[TM_GS_GC_IN_MOVE] = 1;
            gs[TM_GS_GC_PHASE] = vreg((seq << 4) | GC_DONE);
        }
    };
    if (tid < n_list && M.list_parts[tid] != 0 && M.list_part[tid] < 0 && !M.list_mark[tid]) arrive(tid, false);
    const bool marks_in_lds = gc_marks_in_lds(N) && !(flags & TM_SIM_GC_MEM_MARKS);
    for (int e = 0; e < M.n_order; ++e) {
        const int k = __builtin_amdgcn_readfirstlane((int)M.order[e]);
        const int g = __builtin_amdgcn_readfirstlane(M.list_g[k]), ph = __builtin_amdgcn_readfirstlane(M.list_step[k]);
        const int my_part = __builtin_amdgcn_readfirstlane((int)M.list_part[k]), n_parts = __builtin_amdgcn_readfirstlane((int)M.list_parts[k]);
        const long long p0 = my_part < 0 ? 0 : my_part, p1 = my_part < 0 ? 0 : my_part + 1;      // share = [x * p0 / n_parts, x * p1 / n_parts)
        const GP P = game_ptrs(S, g);
        int32_t* gs = P.gs();
        uint32_t* nmw = reinterpret_cast<uint32_t*>(S.gc_mark + (size_t)g * 2 * bm_bytes);
        uint32_t* omw = reinterpret_cast<uint32_t*>(S.gc_mark + (size_t)g * 2 * bm_bytes + bm_bytes);
        int32_t* part = S.gc_part + (size_t)g * TM_GC_PART_DW;
        if (my_part < 0) continue;        // (a marking only: the second loop)
        if (ph == GCP_INIT || ph == GC_SPEC_REQ || ph == GC_REQ_SPEC) {
            if (ph != GC_SPEC_REQ) {          // the tables are cleared once the game has stopped (it looks things up in them)
                uint4* nt4 = reinterpret_cast<uint4*>(P.ntab());
                uint4* ot4 = reinterpret_cast<uint4*>(P.otab());
                const uint4 z4 = make_uint4(0, 0, 0, 0);
                const long long n4 = S.table_cap / 2;
                const int lo = (int)(n4 * p0 / n_parts), hi = (int)(n4 * p1 / n_parts);
                for (int i = lo + tid; i < hi; i += T) { nt4[i] = z4; ot4[i] = z4; }
            }
            if (ph != GC_REQ_SPEC && my_part == g % n_parts) {
                // a marking begins with the next launch (a speculative one: when the game's write barrier is up): empty bitmaps,
                // the root pending, its chunk flagged
                uint32_t* pdw = reinterpret_cast<uint32_t*>(S.gc_queue + (size_t)g * N);
                for (size_t i = tid; i < 2 * bm_bytes / 4; i += T) nmw[i] = 0;      // both bitmaps
                for (size_t i = tid; i < 2 * bm_bytes / 4; i += T) pdw[i] = 0;      // the pending bits, and what the barrier has sent
                __syncthreads();
                if (tid == 0) {
                    // the root is SENT to the owner of its share (a mark is an owner's to set); the empty tree's root is node 0
                    const int root = gs[TM_GS_ROOT], ck = root >> gc_chunk_log2(N);
                    if (root != 0) pdw[root >> 5] = 1u << (root & 31);
                    else { nmw[0] = 1u; nmw[bm_bytes / 4] = 1u; }
                    for (int i = 0; i < TM_GC_FLAG_WORDS; ++i)
                        atomicExch(&gs[TM_GS_GC_FLAGS + i], (root != 0 && (ck >> 5) == i) ? (int)(1u << (ck & 31)) : 0);
                }
            }
        } else if (ph == GCP_COUNT) {
            const int lo = (int)((long long)n_words * p0 / n_parts), hi = (int)((long long)n_words * p1 / n_parts);
            const int low_obs = gs[TM_GS_LOW_OBS];
            int nf = 0, of = 0, kf = 0;
            for (int wi = lo + tid; wi < hi; wi += T) {
                const int rem = N - wi * 32;
                const uint32_t valid = rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u);
                nf += __popc(~nmw[wi] & valid);
                if (dharvest) kf += dist_mark_word(S, P, ~nmw[wi] & valid, wi, gs[TM_GS_LOW_NODE]);     // (nothing is cleared in this step)
                const uint32_t ofr = ~omw[wi] & valid;
                of += __popc(ofr);
                if (harvest && wi * 32 + 31 >= low_obs) {
                    for (uint32_t bits = ofr; bits; bits &= bits - 1) {
                        const int o = wi * 32 + (__ffs(bits) - 1);
                        const uint32_t x = o >= low_obs ? P.stat()[(size_t)o * 4] : 0u;
                        if ((int)x >= S.min_visits_to_store && !(x >> 31)) kf += 1;
                    }
                }
            }
            int t0, t1, t2;
            G_::exscan(nf, tid, sm, t0);
            G_::exscan(of, tid, sm, t1);
            G_::exscan(kf, tid, sm, t2);
            if (tid == 0 && my_part >= 0) { part[3 * my_part] = t0; part[3 * my_part + 1] = t1; part[3 * my_part + 2] = t2; }
        } else if (ph == GCP_WRITE) {
            // Freed node records are not cleared: new_node initialises a slot completely when it hands it out (the reference
            // zeroes at GC, agents/agent.py:227-244; nothing reads a free slot in between).  Of a freed observation only
            // the statistics are cleared (16 B): a visit count of 0 keeps a slot that stays free from being harvested again.
            const int lo = (int)((long long)n_words * p0 / n_parts), hi = (int)((long long)n_words * p1 / n_parts);
            const int low_obs = gs[TM_GS_LOW_OBS], low_node = gs[TM_GS_LOW_NODE];
            int nbase = 0, obase = 0, kbase = (harvest || dharvest) ? S.replay_count[g] : 0;      // (the count is updated by the step's last workgroup)
            for (int b = 0; b < my_part; ++b) { nbase += part[3 * b]; obase += part[3 * b + 1]; kbase += part[3 * b + 2]; }
            for (int wbase = lo; wbase < hi; wbase += T) {
                const int wi = wbase + tid;
                uint32_t valid = 0;
                if (wi < hi) { const int rem = N - wi * 32; valid = rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u); }
                const uint32_t fr = (wi < hi) ? (~nmw[wi] & valid) : 0u;
                if (dharvest) {
                    // the tuples the count step chose (header bit 26), copied before this thread clears the slots
                    const uint32_t keepmask = dist_marked(P, fr, wi, low_node);
                    int ktotal;
                    int kpos = kbase + G_::exscan(__popc(keepmask), tid, sm, ktotal);
                    kbase += ktotal;
                    for (uint32_t bits = keepmask; bits; bits &= bits - 1) {
                        if (kpos < S.replay_cap) dist_harvest_store(S, P, g, wi * 32 + (__ffs(bits) - 1), kpos);
                        kpos += 1;
                    }
                }
                int total;
                int pos = nbase + G_::exscan(__popc(fr), tid, sm, total);
                for (uint32_t bits = fr; bits; bits &= bits - 1) {
                    const int i = wi * 32 + (__ffs(bits) - 1);
                    P.fnode()[pos++] = i;
                    // A free slot holds no child indices and no observation: a speculative marking may visit a node that the
                    // game is creating in the same launch (it finds it in its parent's child row) before the node's own
                    // initialisation is visible - it must not follow what the slot held in its previous life.
                    if (i >= low_node) {
                        const uint4 z4 = make_uint4(0, 0, 0, 0);
                        uint4* kd = reinterpret_cast<uint4*>(P.kids() + (size_t)i * TM_KIDS_DW);
                        kd[0] = z4; kd[1] = z4;
                        reinterpret_cast<uint4*>(P.rec() + (size_t)i * TM_REC_DW)[7] = z4;
                    }
                }
                nbase += total;
                const uint32_t ofr = (wi < hi) ? (~omw[wi] & valid) : 0u;
                int ototal;
                int opos = obase + G_::exscan(__popc(ofr), tid, sm, ototal);
                obase += ototal;
                if (harvest) {
                    // store_nodes (ValueSim.py:122-159): freed observations with enough visits that are not terminal
                    uint32_t keepmask = 0;
                    if (wi < hi && wi * 32 + 31 >= low_obs)
                        for (uint32_t bits = ofr; bits; bits &= bits - 1) {
                            const int o = wi * 32 + (__ffs(bits) - 1);
                            const uint32_t x = o >= low_obs ? P.stat()[(size_t)o * 4] : 0u;
                            if ((int)x >= S.min_visits_to_store && !(x >> 31)) keepmask |= 1u << (o & 31);
                        }
                    int ktotal;
                    int kpos = kbase + G_::exscan(__popc(keepmask), tid, sm, ktotal);
                    kbase += ktotal;
                    for (uint32_t bits = keepmask; bits; bits &= bits - 1) {
                        const int o = wi * 32 + (__ffs(bits) - 1);
                        if (kpos < S.replay_cap) {
                            const uint4 st = *reinterpret_cast<const uint4*>(P.stat() + (size_t)o * 4);
                            const uint4* sk = reinterpret_cast<const uint4*>(P.okey() + (size_t)o * TM_OBS_DW);
                            uint4* dk = reinterpret_cast<uint4*>(S.replay_obs + ((size_t)g * S.replay_cap + kpos) * TM_OBS_DW);
                            dk[0] = sk[0]; dk[1] = sk[1]; dk[2] = sk[2];
                            float* ds = S.replay_stat + ((size_t)g * S.replay_cap + kpos) * 4;
                            ds[0] = __uint_as_float(st.y); ds[1] = __uint_as_float(st.z); ds[2] = (float)(int)st.x; ds[3] = 0.f;
                        }
                        kpos += 1;
                    }
                }
                // (a thread clears only statistics it has read itself above: program order)
                for (uint32_t bits = ofr; bits; bits &= bits - 1) {
                    const int o = wi * 32 + (__ffs(bits) - 1);
                    P.fobs()[opos++] = o;
                    if (o >= low_obs) *reinterpret_cast<uint4*>(P.stat() + (size_t)o * 4) = make_uint4(0, 0, 0, 0);
                }
            }
        }
        // The re-insertions ride along (r06): they read what the steps above do not write - the KEPT nodes' games and the cleared
        // table with the count, the KEPT observations' keys with the write step (which clears and lists the FREED ones) - so a
        // collection is two launches after its marking instead of four.
        if (ph == GCP_COUNT) {
            // the kept nodes, by index (the bitmap says which), back into their (cleared) table: threads claim empty slots with a
            // 64-bit compare-and-swap (no deletions happen concurrently, so linear probing stays consistent; placement order
            // does not affect lookups)
            constexpr int UR = 2;
            const int lo = (int)((long long)N * p0 / n_parts), hi = (int)((long long)N * p1 / n_parts);
            for (int base = lo; base < hi; base += T * UR) {
                uint4 key[UR][4];
                bool kept[UR];
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const int i = base + u * T + tid;
                    kept[u] = i > 0 && i < hi && ((nmw[i >> 5] >> (i & 31)) & 1u);
                    const uint4* src = reinterpret_cast<const uint4*>(P.game() + (size_t)(kept[u] ? i : 0) * GAME_DW);
#pragma unroll
                    for (int t = 0; t < 4; ++t) key[u][t] = src[t];
                }
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    if (!kept[u]) continue;
                    const int i = base + u * T + tid;
                    uint32_t kw[GAME_DW];
#pragma unroll
                    for (int t = 0; t < 4; ++t) { kw[4*t] = key[u][t].x; kw[4*t+1] = key[u][t].y; kw[4*t+2] = key[u][t].z; kw[4*t+3] = key[u][t].w; }
                    const uint64_t h = hash_game(kw);
                    uint32_t sl = (uint32_t)h & mask;
                    const unsigned long long ent = ((h >> 32) << 32) | (uint32_t)i;
                    while (atomicCAS(reinterpret_cast<unsigned long long*>(&P.ntab()[sl]), 0ull, ent) != 0ull) sl = (sl + 1) & mask;
                }
            }
        }
        if (ph == GCP_WRITE && S.kind != TM_KIND_DIST) {    // (no observation table without the projection)
            constexpr int UR = 2;
            {
                const int lo = (int)((long long)N * p0 / n_parts), hi = (int)((long long)N * p1 / n_parts);
                for (int base = lo; base < hi; base += T * UR) {
                    uint4 key[UR][3];
                    bool kept[UR];
#pragma unroll
                    for (int u = 0; u < UR; ++u) {
                        const int o = base + u * T + tid;
                        kept[u] = o > 0 && o < hi && ((omw[o >> 5] >> (o & 31)) & 1u);
                        const uint4* src = reinterpret_cast<const uint4*>(P.okey() + (size_t)(kept[u] ? o : 0) * OBS_DW);
#pragma unroll
                        for (int t = 0; t < 3; ++t) key[u][t] = src[t];
                    }
#pragma unroll
                    for (int u = 0; u < UR; ++u) {
                        if (!kept[u]) continue;
                        const int o = base + u * T + tid;
                        uint32_t kw[OBS_DW];
#pragma unroll
                        for (int t = 0; t < 3; ++t) { kw[4*t] = key[u][t].x; kw[4*t+1] = key[u][t].y; kw[4*t+2] = key[u][t].z; kw[4*t+3] = key[u][t].w; }
                        const uint64_t h = hash_obs(kw);
                        uint32_t sl = (uint32_t)h & mask;
                        const unsigned long long ent = ((h >> 32) << 32) | (uint32_t)o;
                        while (atomicCAS(reinterpret_cast<unsigned long long*>(&P.otab()[sl]), 0ull, ent) != 0ull) sl = (sl + 1) & mask;
                    }
                }
            }
        }
        // ---- arrive (a game this workgroup also marks for - a lone collector workgroup - arrives after the marking) ----
        __syncthreads();
        if (tid == 0 && !M.list_mark[k]) arrive(k, false);
        __syncthreads();
    }
    // the markings of this workgroup (a loop of its own: what a bounded share keeps in registers is dead here)
    for (int e = 0; e < M.n_order; ++e) {
        // (what comes out of LDS is the same for every lane - say so: the game's base pointers live in scalar registers)
        const int k = __builtin_amdgcn_readfirstlane((int)M.order[e]);
        if (!M.list_mark[k]) continue;
        const GP P = game_ptrs(S, __builtin_amdgcn_readfirstlane(M.list_g[k]));
        const long long my_deadline = deadline;      // (a shared workgroup: one game after the other, see the plan)
        // the deadline has come before this game's turn: it stays as it is (and its marking workgroup says so when it arrives)
        if (G_::bcast((deadline >= 0 && (long long)__builtin_readcyclecounter() > deadline) ? 1 : 0, tid, sm)) {
            if (tid == 0) M.list_share[k] = (short)-2;
            continue;
        }
        const bool spec = (__builtin_amdgcn_readfirstlane(M.list_ph[k]) & 15) == GC_SPEC_MARK;
        const int mp = __builtin_amdgcn_readfirstlane((int)M.list_mpart[k]), mps = __builtin_amdgcn_readfirstlane((int)M.list_mparts[k]);
        if (tid == 0 && mp == 0) { P.gs()[TM_GS_GC_MARK_PARTS] += mps; P.gs()[TM_GS_GC_MARK_SHARED] += M.list_share[k]; }
        __syncthreads();
        const bool leftover = marks_in_lds ? gc_sweep_mark<true>(S, P, M, M.marks, mp, mps, spec, my_deadline, tid, sm)
                                           : gc_sweep_mark<false>(S, P, M, nullptr, mp, mps, spec, my_deadline, tid, sm);
        if (tid == 0) M.list_share[k] = (short)(leftover ? -2 : -1);      // (what this workgroup says when it arrives: below)
    }
    // ... and its arrivals for them (a loop of its own: the values the ends of the steps store are not the marker's to carry)
    __syncthreads();
    if (tid == 0)
        for (int e = 0; e < M.n_order; ++e) {
            const int k = M.order[e];
            if (M.list_mark[k]) arrive(k, M.list_share[k] == -2);
        }
}

// ---------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------
// LDS of a k_sim_step workgroup: a simulation workgroup's four WaveLds (+ the four Mersenne twisters of the rollout kinds), or a
// collector workgroup's GcLds + the two mark bitmaps of the game it marks - one dynamic block, sized for the larger of the two
__host__ inline size_t sim_lds_bytes(const tm_store& S, bool vanilla) {
    const size_t sim = WPB * sizeof(WaveLds) + (vanilla ? WPB * sizeof(MtLds) : 0), gc = gc_lds_bytes(S.max_nodes);
    return (sim > gc ? sim : gc);
}
template <bool VANILLA>
__global__ __launch_bounds__(64 * WPB, 5) void k_sim_step(tm_store S, int flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sim_lds[];
    WaveLds* lds = reinterpret_cast<WaveLds*>(sim_lds);
    MtLds* mt_lds = reinterpret_cast<MtLds*>(sim_lds + WPB * sizeof(WaveLds));   // WPB entries, only in the VANILLA instantiation
    // the wave index is uniform by construction: say so, and every per-game base pointer lives in scalar registers
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // The first workgroups of the grid are collectors: each looks after the garbage collections of a range of games, in
    // slices of S.gc_slice_cycles per launch (catch-up launches pass TM_SIM_GC_FULL: to completion), beside the
    // simulation workgroups - they are dispatched first and share the CUs with them (a k_sim_step wave needs 70 registers).
    const int n_gc = gc_blocks(S);
    if ((int)blockIdx.x < n_gc) {
        // the dense request list: this launch appends under S.eval_parity; the other set of counters (the list the evaluator
        // drew from before this launch) is cleared for the next one
        if (blockIdx.x == 0 && (flags & TM_SIM_FRONT) && (int)threadIdx.x < TM_EVAL_SEGS(S.n_games))
            S.eval_cnt[(size_t)threadIdx.x * 2 + (S.eval_parity ^ 1)] = 0;
        // (a collector wave shares its SIMD with four simulation waves and is the one a blocked game waits for: it goes first)
        __builtin_amdgcn_s_setprio(3);
#ifdef TM_TIMELINE   // (measurement builds, scripts/launch_timeline.py: when this workgroup started and ended, 100 MHz ticks)
        const unsigned tl0 = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
        gc_collector_block(S, flags, n_gc, *reinterpret_cast<GcLds*>(sim_lds));
#ifdef TM_TIMELINE
        if (lane == 0 && (flags & TM_SIM_FRONT) && !(flags & TM_SIM_GC_FULL) && !S.game_list && (int)blockIdx.x + 6 * n_gc < S.n_games) {
            // collector b: the spare word 35 of game b's block (start), of game n_gc + b's (end), of game 2 n_gc + b's (the
            // launch); odd launches 3 n_gc further on
            const size_t row = (size_t)blockIdx.x + ((((unsigned)flags >> 8) & 1) ? 3 * n_gc : 0);
            if (w == 0) S.gs[row * TM_GS_DW + TM_GS_GC_RSV1] = (int)tl0;
            if (w == 0) S.gs[(row + 2 * n_gc) * TM_GS_DW + TM_GS_GC_RSV1] = (int)((unsigned)flags >> 8);
            // (the end as launch << 16 | ticks since the start, the largest over the workgroup's waves: grows from launch to launch
            // whatever the 32-bit clock does - good for 65 535 launches)
            const unsigned dt = (unsigned)__builtin_amdgcn_s_memrealtime() - tl0;
            atomicMax(reinterpret_cast<unsigned*>(S.gs + (row + n_gc) * TM_GS_DW + TM_GS_GC_RSV1), ((((unsigned)flags >> 8) & 0xFFFFu) << 16) | (dt < 0xFFFFu ? dt : 0xFFFFu));
        }
#endif
        return;
    }
    // simulation wave i takes game i, or - a launch over some of the games: the catch-up launches - game_list[i]
    const int slot = ((int)blockIdx.x - n_gc) * WPB + w;
    if (slot >= (S.game_list ? S.n_listed : S.n_games)) return;
    const int g = S.game_list ? __builtin_amdgcn_readfirstlane(S.game_list[slot]) : slot;
    GP P = game_ptrs(S, g);
    WaveLds& L = lds[w];
    int32_t* gs = P.gs();
#ifdef TM_TIMELINE
    const unsigned tl0 = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
    // The game's control block (64 words) in one coalesced load: word i in lane i, read with v_readlane.  A word is read
    // from the snapshot only before this launch writes it (the halves below write each word once, from lane 0).
    int gsv = gs[lane];
    int gc_req_word = (int)(((unsigned)flags >> 8) << 4) | GC_REQ;
    {
        // a collection requested, in progress, or completed in THIS launch: the game does not simulate (what a collector
        // workgroup of this launch writes is not ordered with this wave's reads); completed in an earlier launch: resume;
        // a speculative marking (GC_SPEC_*): the game simulates, with the write barrier once the marking is under way
        const int gcw = GSV(gsv, TM_GS_GC_PHASE), gph = gcw & 15;
        if (gph == GC_SPEC_MARK) gc_req_word = (gc_req_word & ~15) | GC_REQ_SPEC;
        else if (gph == GC_SPEC_REQ) gc_req_word = (gc_req_word & ~15) | GC_REQ_OVER;
        else if (gcw != 0) {
            if (gph != GC_DONE || (gcw >> 4) == (int)((unsigned)flags >> 8)) return;
            if (lane == 0) { gs[TM_GS_GC_PHASE] = 0; gc_active_clear(P); }
            gsv = lane == TM_GS_GC_PHASE ? 0 : gsv;       // (the snapshot's phase word is what the finish stage tests)
        }
    }
    const int pend = GSV(gsv, TM_GS_PENDING);
    if (pend == 2) {
        // the simulation suspended in its expansion: redo the expansion of its leaf, post the evaluation requests
        const int leaf = GSV(gsv, TM_GS_LEAF);
        const uint32_t self_o = P.rec()[(size_t)leaf * TM_REC_DW + TM_REC_OBS];
        const uint32_t self_sc = P.rec()[(size_t)leaf * TM_REC_DW + TM_REC_SCORE];
        wave_front_finish<VANILLA>(S, P, L, g, lane, leaf, 0, self_o, self_sc, gsv, gc_req_word, flags);
        return;
    }
    const bool dist = !VANILLA && S.kind == TM_KIND_DIST;
    const long long t0 = __builtin_readcyclecounter();
    if ((flags & TM_SIM_BACKUP) && pend == 1) {
        if (dist) wave_dist_back(S, P, L, g, lane, gsv);
        else wave_sim_back(S, P, L, lane, gsv, flags);
        __threadfence_block();
    }
    const long long t1 = __builtin_readcyclecounter();
#ifndef TM_TIMELINE
    if (lane == 0) gs[TM_GS_CYC_BACK] = (int)(t1 - t0);
#endif
    // the per-move quota (tm_move_begin): games that lost launches to a collection catch up in extra launches
    if ((flags & TM_SIM_FRONT) && GSV(gsv, TM_GS_SIM_STARTED) < GSV(gsv, TM_GS_SIM_TARGET)) {
        if (dist) wave_dist_front<VANILLA>(S, P, L, g, lane, gsv, gc_req_word, flags);
        else wave_sim_front<VANILLA>(S, P, L, VANILLA ? &mt_lds[w] : nullptr, g, lane, gsv, gc_req_word, flags);
    }
    else if (lane < S.eval_slots) P.eval_obs()[lane] = 0;   // nothing started: no request (the evaluator skips empty slots)
#ifdef TM_TIMELINE
    if (lane == 0 && (flags & TM_SIM_FRONT) && !(flags & TM_SIM_GC_FULL) && !S.game_list &&
        GSV(gsv, TM_GS_SIM_STARTED) < GSV(gsv, TM_GS_SIM_TARGET)) {      // (the move's regular launches: all games; a simulation was started)
        gs[TM_GS_CYC_BACK] = (int)tl0;            // (in place of the phase cycles, which nothing on the device reads)
        gs[TM_GS_CYC_SELECT] = (int)__builtin_amdgcn_s_memrealtime();
        gs[TM_GS_CYC_EXPAND] = (int)((unsigned)flags >> 8);
    }
#endif
}

// per-move simulation quota (TreeAgent.play: self.mcts(self.root, self.sims), agents/agent.py:147-150)
__global__ void k_move_begin(tm_store S, int sims) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= S.n_games) return;
    int32_t* gs = S.gs + (size_t)g * TM_GS_DW;
    gs[TM_GS_SIM_TARGET] = gs[TM_GS_SIM_STARTED] + sims;
    // no request is pending between two moves: both sets of the dense list's counters start the move empty, whatever parity
    // the caller's launches go on with
    S.eval_cnt[(size_t)g * 2] = 0;
    S.eval_cnt[(size_t)g * 2 + 1] = 0;
}
// launches still needed before every game has finished its quota: max over games of (simulations not started yet
// + one launch for the pending backup + one if a collection is in progress); atomicMax into *out (zeroed by the caller)
__global__ void k_sims_remaining(tm_store S, int32_t* out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    int r = 0, col = 0;
    if (g < S.n_games) {
        const int32_t* gs = S.gs + (size_t)g * TM_GS_DW;
        const int ph = gs[TM_GS_GC_PHASE] & 15;
        col = (ph != 0 && ph != GC_DONE && ph != GC_SPEC_REQ && ph != GC_SPEC_MARK) ? 1 : 0;
        r = (gs[TM_GS_SIM_TARGET] - gs[TM_GS_SIM_STARTED]) + (gs[TM_GS_PENDING] != 0 ? 1 : 0) + col;
        // (a game whose collection these launches will not touch - TM_ERR_GC_GRID - never finishes: nobody waits for it,
        // the caller finds the flag)
        if (gs[TM_GS_ERR] & TM_ERR_GC_GRID) { r = 0; col = 0; }
    }
    for (int d = 32; d >= 1; d >>= 1) { r = max(r, __shfl_xor(r, d, 64)); col += __shfl_xor(col, d, 64); }
    if ((threadIdx.x & 63) == 0) { if (r > 0) atomicMax(out, r); if (col > 0) atomicAdd(out + 1, col); }
}

// the same, and the games that still need launches appended to `list` (out[2] = how many; zeroed by the caller)
__global__ void k_sims_owing(tm_store S, int32_t* out, int32_t* list) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    int r = 0, col = 0;
    if (g < S.n_games) {
        const int32_t* gs = S.gs + (size_t)g * TM_GS_DW;
        const int ph = gs[TM_GS_GC_PHASE] & 15;
        col = (ph != 0 && ph != GC_DONE && ph != GC_SPEC_REQ && ph != GC_SPEC_MARK) ? 1 : 0;
        r = (gs[TM_GS_SIM_TARGET] - gs[TM_GS_SIM_STARTED]) + (gs[TM_GS_PENDING] != 0 ? 1 : 0) + col;
        // (a game whose collection these launches will not touch - TM_ERR_GC_GRID - never finishes: nobody waits for it,
        // the caller finds the flag)
        if (gs[TM_GS_ERR] & TM_ERR_GC_GRID) { r = 0; col = 0; }
    }
    const uint64_t owing = __ballot(r > 0);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && owing != 0ull) base = atomicAdd(out + 2, __popcll(owing));
    base = __shfl(base, 0, 64);
    if (r > 0) list[base + __popcll(owing & ((1ull << (threadIdx.x & 63)) - 1ull))] = g;
    for (int d = 32; d >= 1; d >>= 1) { r = max(r, __shfl_xor(r, d, 64)); col += __shfl_xor(col, d, 64); }
    if ((threadIdx.x & 63) == 0) { if (r > 0) atomicMax(out, r); if (col > 0) atomicAdd(out + 1, col); }
}

// agent.update_root(game) (agents/agent.py:296-301)
__global__ __launch_bounds__(64 * WPB) void k_update_root(tm_store S) {
    __shared__ WaveLds lds[WPB];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * WPB + w;
    if (g >= S.n_games) return;
    GP P = game_ptrs(S, g);
    WaveLds& L = lds[w];
    if (lane < GAME_DW) L.slots[0][lane] = S.env_game[(size_t)g * GAME_DW + lane];
    gc_drop_speculative(P, lane);      // the root moves: a speculative marking's marks would be stale
    wave_sync();
    int idx, o;
    wave_new_nodes(S, P, L, g, 1, lane, idx, o);
    if (idx < 0) {
        gc_wave(S, P, g, lane);   // reachable set of the OLD root, as in the reference
        wave_new_nodes(S, P, L, g, 1, lane, idx, o);
        if (idx < 0) { if (lane == 0) atomicOr(&P.gs()[TM_GS_ERR], TM_ERR_POOL); idx = 0; }
    }
    if (lane == 0) {
        P.gs()[TM_GS_ROOT] = idx;
        P.gs()[TM_GS_POOL_FULL] = 0;        // a new root: the old root's siblings are garbage now (TM_GS_GC_IN_MOVE was cleared above)
        if ((L.slots[0][11] >> 8) & 1u) P.gs()[TM_GS_EPISODE] += 1;
    }
}

// TreeAgent's single calls, one game state per tree (agent.cpp:212-218 expand(buffer), 265-270 new_node(buffer);
// agents/agent.py:90-145): idx = new_node(game), and for EXPAND the seven successors linked under it.  As in the
// reference a collection may run at any of the pops; a game that is not reachable from the root does not survive it.
template <bool EXPAND>
__global__ __launch_bounds__(64 * WPB) void k_tree_node(tm_store S, const uint32_t* __restrict__ games,
                                                        const uint8_t* __restrict__ mask, int32_t* __restrict__ out_idx) {
    __shared__ WaveLds lds[WPB];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int g = blockIdx.x * WPB + w;
    if (g >= S.n_games) return;
    if (mask && !mask[g]) { if (out_idx && lane == 0) out_idx[g] = -1; return; }
    GP P = game_ptrs(S, g);
    WaveLds& L = lds[w];
    if (lane < GAME_DW) L.slots[0][lane] = games[(size_t)g * GAME_DW + lane];
    gc_drop_speculative(P, lane);      // the tree changes outside a simulation launch: no barrier watches it
    wave_sync();
    int idx, o;
    wave_new_nodes(S, P, L, g, 1, lane, idx, o);
    if (idx < 0) {
        gc_wave(S, P, g, lane);
        wave_new_nodes(S, P, L, g, 1, lane, idx, o);
        if (idx < 0) { if (lane == 0) atomicOr(&P.gs()[TM_GS_ERR], TM_ERR_POOL); idx = 0; }
    }
    idx = (int)rl_u32((uint32_t)idx, 0);
    if (out_idx && lane == 0) out_idx[g] = idx;
    if (!EXPAND) return;
    __threadfence_block();
    for (int attempt = 0; attempt < 2; ++attempt) {
        const uint32_t self_sc = P.rec()[(size_t)idx * TM_REC_DW + TM_REC_SCORE];
        const int gsv = P.gs()[lane];
        uint32_t lh;
        if (wave_expand(S, P, L, g, lane, idx, self_sc, lh, gsv, GC_REQ)) break;
        // the pool ran dry at one of the seven pops: the collection wave_expand asked for, now, then the expansion again
        // (the successors already inserted are transposition hits; a second failure is TM_ERR_POOL inside wave_expand)
        __threadfence_block();
        gc_collect<64>(S, P, g, lane, nullptr);
    }
    if (lane == 0) { P.gs()[TM_GS_GC_RETRY] = 0; P.gs()[TM_GS_N_EXPAND] += 1; }
}
// TreeAgent.remove_nodes() (agent.cpp:337-..., agents/agent.py:246-257) as a call of its own: one workgroup per game
__global__ __launch_bounds__(64 * WPB) void k_tree_gc(tm_store S, const uint8_t* __restrict__ mask) {
    __shared__ int sm[16];
    const int g = blockIdx.x;
    if (mask && !mask[g]) return;
    GP P = game_ptrs(S, g);
    if (threadIdx.x == 0) { P.gs()[TM_GS_GC_PHASE] = GC_REQ; gc_active_set(P); P.gs()[TM_GS_GC_IN_MOVE] = 0; }      // (a speculative marking is simply overwritten)
    __syncthreads();
    gc_collect<64 * WPB>(S, P, g, (int)threadIdx.x, sm);
}

// compute_stats + get_action (agents/agent.py:153-185; agent.cpp:149-172 for the all-C++ kinds)
__global__ void k_root_stats(tm_store S, float* stats, int32_t* action) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= S.n_games) return;
    GP P = game_ptrs(S, g);
    const int root = P.gs()[TM_GS_ROOT];
    float self = __uint_as_float(P.rec()[(size_t)root * TM_REC_DW + TM_REC_SCORE]);
    float* out = stats + (size_t)g * 21;
    if (S.kind == TM_KIND_DIST) {
        // DistValueSimOnline.py:76-98: visit / multiplicity of the child among the seven actions, mean + score - root score
        // (left to right, float), variance; the action = np.argmax of the second row
        int bestd = 0, nan_at = -1;
        float bestq = 0;
        for (int a = 0; a < 7; ++a) {
            const uint32_t c = P.kids()[(size_t)root * TM_KIDS_DW + a];
            int cnt = 0;
            for (int b = 0; b < 7; ++b) cnt += P.kids()[(size_t)root * TM_KIDS_DW + b] == c;
            const float sc = __uint_as_float(P.rec()[(size_t)c * TM_REC_DW + TM_REC_SCORE]);
            const uint4 st = *reinterpret_cast<const uint4*>(P.stat() + (size_t)c * 4);
            const float t = __uint_as_float(st.y) + sc;
            const float v1 = t - self;
            out[a] = __uint_as_float(st.x) / (float)cnt;
            out[7 + a] = v1;
            out[14 + a] = __uint_as_float(st.z);
            if (v1 != v1 && nan_at < 0) nan_at = a;
            if (a == 0 || v1 > bestq) { bestq = v1; bestd = a; }
        }
        action[g] = nan_at >= 0 ? nan_at : bestd;
        return;
    }
    const bool cpp = (S.kind == TM_KIND_CPPAGENT_LP || S.kind == TM_KIND_CPPAGENT || S.kind == TM_KIND_VANILLA_C);
    int best = 0, first_nan = -1;
    float bestv = 0;
    for (int a = 0; a < 7; ++a) {
        const uint32_t c = P.kids()[(size_t)root * TM_KIDS_DW + a];    // child 0 = null node: record all zero
        const uint32_t o = P.rec()[(size_t)c * TM_REC_DW + TM_REC_OBS];
        const float sc = __uint_as_float(P.rec()[(size_t)c * TM_REC_DW + TM_REC_SCORE]);
        uint4 st = *reinterpret_cast<const uint4*>(P.stat() + (size_t)o * 4);
        float value = __uint_as_float(st.y);
        float v1;
        if (cpp) { float t = value + sc; v1 = t - self; }
        else { float diff = sc - self; v1 = value + diff; }
        out[a] = (float)(int)(st.x & 0x7FFFFFFFu);
        out[7 + a] = v1;
        out[14 + a] = __uint_as_float(st.z);
        if (v1 != v1 && first_nan < 0) first_nan = a;
        if (a == 0 || v1 > bestv) { if (a == 0 || v1 > bestv) { bestv = v1; best = a; } }
    }
    // np.argmax: first NaN wins; std::max_element (agent.cpp:170): NaN never compares greater
    action[g] = (!cpp && first_nan >= 0) ? first_nan : best;
}

__global__ void k_pool_init(tm_store S) {
    int g = blockIdx.x;
    GP P = game_ptrs(S, g);
    const int N = S.max_nodes;
    for (int i = threadIdx.x; i < N - 1; i += blockDim.x) { P.fnode()[i] = i + 1; P.fobs()[i] = i + 1; }
    if (threadIdx.x == 0) {
        for (int i = 0; i < TM_GS_DW; ++i) P.gs()[i] = 0;
        P.gs()[TM_GS_NFREE_NODE] = N - 1;
        P.gs()[TM_GS_NFREE_OBS] = N - 1;
        P.gs()[TM_GS_LOW_NODE] = N;
        P.gs()[TM_GS_LOW_OBS] = N;
        uint32_t r[32];
        int pos;
        srand_state(r, pos, 1u);   // the reference never calls srand(): glibc's default seed
        for (int i = 0; i < 32; ++i) S.rng[(size_t)g * 32 + i] = r[i];
        P.gs()[TM_GS_RNG_POS] = pos;
        if (S.replay_count) S.replay_count[g] = 0;
    }
}

// Re-initialise the trees of the games selected by mask (pool exhausted beyond what GC can reclaim: the reference has
// undefined behaviour there, agent.cpp:227-231).  Episode counter, rand() stream and statistics counters survive.
__global__ void k_pool_reset(tm_store S, const uint8_t* mask) {
    const int g = blockIdx.x;
    if (!mask[g]) return;
    GP P = game_ptrs(S, g);
    const int N = S.max_nodes;
    // (a selected game's 40 MB are cleared by gridDim.y workgroups: one took 0.4 ms, a move's worth of 22 games under the trained net)
    const size_t t0 = (size_t)blockIdx.y * blockDim.x + threadIdx.x, stride = (size_t)gridDim.y * blockDim.x;
    const uint4 z = make_uint4(0, 0, 0, 0);
    uint4* rec = reinterpret_cast<uint4*>(P.rec());
    for (size_t i = t0; i < (size_t)N * TM_REC_DW / 4; i += stride) rec[i] = z;
    uint4* kd = reinterpret_cast<uint4*>(P.kids());
    for (size_t i = t0; i < (size_t)N * TM_KIDS_DW / 4; i += stride) kd[i] = z;
    uint4* gm = reinterpret_cast<uint4*>(P.game());
    for (size_t i = t0; i < (size_t)N * GAME_DW / 4; i += stride) gm[i] = z;
    uint4* stt = reinterpret_cast<uint4*>(P.stat());
    for (size_t i = t0; i < (size_t)N; i += stride) stt[i] = z;
    uint4* ok = reinterpret_cast<uint4*>(P.okey());
    for (size_t i = t0; i < (size_t)N * OBS_DW / 4; i += stride) ok[i] = z;
    for (size_t i = t0; i < (size_t)S.table_cap; i += stride) { P.ntab()[i] = 0; P.otab()[i] = 0; }
    for (size_t i = t0; i < (size_t)N - 1; i += stride) { P.fnode()[i] = (int)i + 1; P.fobs()[i] = (int)i + 1; }
    if (t0 == 0) {
        P.gs()[TM_GS_ROOT] = 0;
        P.gs()[TM_GS_NFREE_NODE] = N - 1;
        P.gs()[TM_GS_NFREE_OBS] = N - 1;
        P.gs()[TM_GS_LOW_NODE] = N;
        P.gs()[TM_GS_LOW_OBS] = N;
        P.gs()[TM_GS_TRACE_LEN] = 0;
        P.gs()[TM_GS_PENDING] = 0;
        P.gs()[TM_GS_GC_PHASE] = 0;
        gc_active_clear(P);
        P.gs()[TM_GS_GC_RETRY] = 0;
        P.gs()[TM_GS_POOL_FULL] = 0;
        P.gs()[TM_GS_GC_IN_MOVE] = 0;
        P.gs()[TM_GS_GC_ARRIVE] = 0;
        P.gs()[TM_GS_SIM_STARTED] = P.gs()[TM_GS_SIM_TARGET];
        P.gs()[TM_GS_ERR] &= ~TM_ERR_POOL;
        P.gs()[TM_GS_N_POOL_RESET] += 1;
    }
}

// ---- environment kernels: one lane per game, 64-byte LDS slot per lane ----
__global__ __launch_bounds__(64) void k_env_init(tm_store S, const uint32_t* seeds) {
    __shared__ uint32_t slots[64][GAME_DW];
    int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= S.n_games) return;
    EngCfg cfg{S.app, S.scoring, S.randomizer};
    init_game(slots[threadIdx.x], cfg, seeds[g]);
    for (int i = 0; i < GAME_DW; ++i) S.env_game[(size_t)g * GAME_DW + i] = slots[threadIdx.x][i];
    for (int i = 0; i < 4; ++i) S.env_line_stats[(size_t)g * 4 + i] = 0;
}
__global__ __launch_bounds__(64) void k_env_step(tm_store S, const int32_t* actions) {
    __shared__ uint32_t slots[64][GAME_DW];
    int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= S.n_games) return;
    uint32_t* sl = slots[threadIdx.x];
    for (int i = 0; i < GAME_DW; ++i) sl[i] = S.env_game[(size_t)g * GAME_DW + i];
    EngCfg cfg{S.app, S.scoring, S.randomizer};
    int ls[4];
    for (int i = 0; i < 4; ++i) ls[i] = S.env_line_stats[(size_t)g * 4 + i];
    Piece p;
    load_fields(sl, p);
    // line_stats is a tiny fixed array: count through a local copy
    int before = p.line_clears;
    play(reinterpret_cast<uint16_t*>(sl), p, cfg, actions[g], nullptr);
    int n = p.line_clears - before;
    if (n > 0) ls[n - 1] += 1;
    store_fields(sl, p);
    for (int i = 0; i < GAME_DW; ++i) S.env_game[(size_t)g * GAME_DW + i] = sl[i];
    for (int i = 0; i < 4; ++i) S.env_line_stats[(size_t)g * 4 + i] = ls[i];
}
__global__ __launch_bounds__(64) void k_env_reset(tm_store S, const uint8_t* mask) {
    __shared__ uint32_t slots[64][GAME_DW];
    int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= S.n_games) return;
    uint32_t flags = (S.env_game[(size_t)g * GAME_DW + 11] >> 8) & 0xFF;
    bool doit = mask ? (mask[g] != 0) : ((flags & 1u) != 0);
    if (!doit) return;
    EngCfg cfg{S.app, S.scoring, S.randomizer};
    uint32_t seed = rnd32(S.env_game[(size_t)g * GAME_DW + 13], 0xFFFFFFFFu);
    init_game(slots[threadIdx.x], cfg, seed);
    for (int i = 0; i < GAME_DW; ++i) S.env_game[(size_t)g * GAME_DW + i] = slots[threadIdx.x][i];
    for (int i = 0; i < 4; ++i) S.env_line_stats[(size_t)g * 4 + i] = 0;
}
__global__ void k_env_render(tm_store S, int8_t* out) {
    __shared__ uint32_t key[OBS_DW];
    __shared__ uint32_t gm[GAME_DW];
    int g = blockIdx.x;
    if (threadIdx.x < GAME_DW) gm[threadIdx.x] = S.env_game[(size_t)g * GAME_DW + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) pack_obs(gm, key);
    __syncthreads();
    if (threadIdx.x < 200) out[(size_t)g * 200 + threadIdx.x] = (int8_t)obs_cell(key, threadIdx.x);
}
__global__ void k_env_info(tm_store S, int32_t* out) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= S.n_games) return;
    const uint32_t* e = S.env_game + (size_t)g * GAME_DW;
    out[g * 8 + 0] = (e[11] >> 8) & 1;
    out[g * 8 + 1] = (int)e[14];
    out[g * 8 + 2] = (int)e[15];
    out[g * 8 + 3] = (int)(int16_t)(e[11] >> 16);
    for (int i = 0; i < 4; ++i) out[g * 8 + 4 + i] = S.env_line_stats[(size_t)g * 4 + i];
}

// evaluation requests -> int8 [G*K][200] observations (evaluator type 0 input, agent.cpp:424-436)
__global__ void k_eval_render(tm_store S, int8_t* out) {
    __shared__ uint32_t key[OBS_DW];
    int j = blockIdx.x;
    int g = j / S.eval_slots;
    int o = S.eval_obs[j];
    if (S.kind == TM_KIND_DIST) {      // the request names a node: its observation is packed from its game
        __shared__ uint32_t gm[GAME_DW];
        if (threadIdx.x < GAME_DW) gm[threadIdx.x] = S.node_game[((size_t)g * S.max_nodes + o) * GAME_DW + threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) pack_obs(gm, key);
    } else if (threadIdx.x < OBS_DW) key[threadIdx.x] = S.obs_key[((size_t)g * S.max_nodes + o) * OBS_DW + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 200) out[(size_t)j * 200 + threadIdx.x] = (o == 0) ? (int8_t)0 : (int8_t)obs_cell(key, threadIdx.x);
}

__global__ void k_export_game(tm_store S, int g, int32_t* child, float* score, int32_t* n_to_o, int32_t* visit,
                              float* value, float* variance, uint8_t* end_obs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.max_nodes) return;
    GP P = game_ptrs(S, g);
    const uint32_t* r = P.rec() + (size_t)i * TM_REC_DW;
    for (int a = 0; a < 7; ++a) child[(size_t)i * 7 + a] = (int)P.kids()[(size_t)i * TM_KIDS_DW + a];
    score[i] = __uint_as_float(r[TM_REC_SCORE]);
    n_to_o[i] = (int)r[TM_REC_OBS];
    uint4 st = *reinterpret_cast<const uint4*>(P.stat() + (size_t)i * 4);
    visit[i] = (int)(st.x & 0x7FFFFFFFu);
    value[i] = __uint_as_float(st.y);
    variance[i] = __uint_as_float(st.z);
    end_obs[i] = (uint8_t)(st.x >> 31);
}

}  // namespace tmcts

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
using namespace tmcts;
#define TM_LAUNCH_CHECK() ((int)hipGetLastError())
static std::atomic<unsigned> tm_launch_seq{0};

// TM_GC_MARKS_IN_MEMORY=1 in the environment: every launch's collectors keep their bitmaps in memory (what pools of more than
// 100 000 nodes get) - so that a test can hold that form to the oracle on a pool the oracle replays in seconds
static int gc_mem_marks_flag() {
    const char* e = getenv("TM_GC_MARKS_IN_MEMORY");
    return (e && e[0] == '1') ? TM_SIM_GC_MEM_MARKS : 0;
}

extern "C" {

const char* tm_version(void) { return "tetris_mcts_hip 0.1 (gfx950)"; }

void tm_fill_norm_quantile(float* t, int n) {
    // special.h:26-33 with the host libm, exactly as the reference evaluates it
    const double log2_ = log(2.0), log22 = log(22.0), log41 = log(41.0);
    for (int i = 0; i < n; ++i) {
        double alpha = 1 - 1 / (double)i;
        t[i] = (float)(10 * log(1 - log(-log(alpha) / log2_) / log22) / log41);
    }
}

void tm_fill_norm_quantile_f64(double* t, int n) {
    const double log2_ = log(2.0), log22 = log(22.0), log41 = log(41.0);
    for (int i = 0; i < n; ++i) {
        double alpha = 1 - 1 / (double)i;
        t[i] = 10 * log(1 - log(-log(alpha) / log2_) / log22) / log41;
    }
}

int tm_pool_init(const tm_store* s, void* stream) {
    hipLaunchKernelGGL(k_pool_init, dim3(s->n_games), dim3(256), 0, (hipStream_t)stream, *s);
    return TM_LAUNCH_CHECK();
}
int tm_pool_reset(const tm_store* s, const uint8_t* mask, void* stream) {
    hipLaunchKernelGGL(k_pool_reset, dim3(s->n_games, 16), dim3(256), 0, (hipStream_t)stream, *s, mask);
    return TM_LAUNCH_CHECK();
}
int tm_env_init(const tm_store* s, const uint32_t* seeds, void* stream) {
    hipLaunchKernelGGL(k_env_init, dim3((s->n_games + 63) / 64), dim3(64), 0, (hipStream_t)stream, *s, seeds);
    return TM_LAUNCH_CHECK();
}
int tm_env_step(const tm_store* s, const int32_t* actions, void* stream) {
    hipLaunchKernelGGL(k_env_step, dim3((s->n_games + 63) / 64), dim3(64), 0, (hipStream_t)stream, *s, actions);
    return TM_LAUNCH_CHECK();
}
int tm_env_reset(const tm_store* s, const uint8_t* mask, void* stream) {
    hipLaunchKernelGGL(k_env_reset, dim3((s->n_games + 63) / 64), dim3(64), 0, (hipStream_t)stream, *s, mask);
    return TM_LAUNCH_CHECK();
}
int tm_env_render(const tm_store* s, int8_t* out, void* stream) {
    hipLaunchKernelGGL(k_env_render, dim3(s->n_games), dim3(256), 0, (hipStream_t)stream, *s, out);
    return TM_LAUNCH_CHECK();
}
int tm_env_info(const tm_store* s, int32_t* out, void* stream) {
    hipLaunchKernelGGL(k_env_info, dim3((s->n_games + 255) / 256), dim3(256), 0, (hipStream_t)stream, *s, out);
    return TM_LAUNCH_CHECK();
}
int tm_update_root(const tm_store* s, void* stream) {
    hipLaunchKernelGGL(k_update_root, dim3((s->n_games + WPB - 1) / WPB), dim3(64 * WPB), 0, (hipStream_t)stream, *s);
    return TM_LAUNCH_CHECK();
}
int tm_tree_new_node(const tm_store* s, const uint32_t* games, const uint8_t* mask, int32_t* out_idx, void* stream) {
    if (!games) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_tree_node<false>, dim3((s->n_games + WPB - 1) / WPB), dim3(64 * WPB), 0, (hipStream_t)stream, *s,
                       games, mask, out_idx);
    return TM_LAUNCH_CHECK();
}
int tm_tree_expand(const tm_store* s, const uint32_t* games, const uint8_t* mask, int32_t* out_idx, void* stream) {
    if (!games) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_tree_node<true>, dim3((s->n_games + WPB - 1) / WPB), dim3(64 * WPB), 0, (hipStream_t)stream, *s,
                       games, mask, out_idx);
    return TM_LAUNCH_CHECK();
}
int tm_tree_remove_nodes(const tm_store* s, const uint8_t* mask, void* stream) {
    hipLaunchKernelGGL(k_tree_gc, dim3(s->n_games), dim3(64 * WPB), 0, (hipStream_t)stream, *s, mask);
    return TM_LAUNCH_CHECK();
}
int tm_sim_step(const tm_store* s, int flags, void* stream) {
    if (!s->eval_list || !s->eval_cnt) return (int)hipErrorInvalidValue;      // (a tm_store of an older header: no request list)
    if (s->game_list && (s->n_listed < 0 || s->n_listed > s->n_games)) return (int)hipErrorInvalidValue;
    // the launch number (bits 8.. of the kernel's flags): what orders a game's wave and its collector workgroup, which
    // only ever hand over at kernel boundaries.  Any two launches that touch the same game differ in it.
    flags = (flags & 0xFF) | (int)(((tm_launch_seq.fetch_add(1) + 1u) & 0x7FFFFu) << 8) | gc_mem_marks_flag();
    const int n_waves = s->game_list ? s->n_listed : s->n_games;       // simulation waves; the collectors are those of all n_games
    const dim3 grid((n_waves + WPB - 1) / WPB + gc_blocks(*s)), block(64 * WPB);
    if (s->kind == TM_KIND_VANILLA || s->kind == TM_KIND_VANILLA_C)
        hipLaunchKernelGGL(k_sim_step<true>, grid, block, sim_lds_bytes(*s, true), (hipStream_t)stream, *s, flags);
    else
        hipLaunchKernelGGL(k_sim_step<false>, grid, block, sim_lds_bytes(*s, false), (hipStream_t)stream, *s, flags);
    return TM_LAUNCH_CHECK();
}
int tm_gc_step(const tm_store* s, void* stream) {
    // the collector workgroups alone, no time limit: one step of every collection under way
    const int flags = TM_SIM_GC_FULL | (int)(((tm_launch_seq.fetch_add(1) + 1u) & 0x7FFFFu) << 8) | gc_mem_marks_flag();
    const dim3 grid(gc_blocks(*s)), block(64 * WPB);
    if (s->kind == TM_KIND_VANILLA || s->kind == TM_KIND_VANILLA_C)
        hipLaunchKernelGGL(k_sim_step<true>, grid, block, sim_lds_bytes(*s, true), (hipStream_t)stream, *s, flags);
    else
        hipLaunchKernelGGL(k_sim_step<false>, grid, block, sim_lds_bytes(*s, false), (hipStream_t)stream, *s, flags);
    return TM_LAUNCH_CHECK();
}
int tm_move_begin(const tm_store* s, int sims, void* stream) {
    if (!s->eval_cnt) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_move_begin, dim3((s->n_games + 255) / 256), dim3(256), 0, (hipStream_t)stream, *s, sims);
    return TM_LAUNCH_CHECK();
}
int tm_sims_remaining(const tm_store* s, int32_t* out, void* stream) {
    hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(int32_t), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_sims_remaining, dim3((s->n_games + 255) / 256), dim3(256), 0, (hipStream_t)stream, *s, out);
    return TM_LAUNCH_CHECK();
}
int tm_sims_owing(const tm_store* s, int32_t* out, int32_t* list, void* stream) {
    if (!list) return (int)hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(out, 0, 3 * sizeof(int32_t), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_sims_owing, dim3((s->n_games + 255) / 256), dim3(256), 0, (hipStream_t)stream, *s, out, list);
    return TM_LAUNCH_CHECK();
}
int tm_eval_render(const tm_store* s, int8_t* out, void* stream) {
    hipLaunchKernelGGL(k_eval_render, dim3(s->n_games * s->eval_slots), dim3(256), 0, (hipStream_t)stream, *s, out);
    return TM_LAUNCH_CHECK();
}
int tm_root_stats(const tm_store* s, float* stats, int32_t* action, void* stream) {
    hipLaunchKernelGGL(k_root_stats, dim3((s->n_games + 255) / 256), dim3(256), 0, (hipStream_t)stream, *s, stats, action);
    return TM_LAUNCH_CHECK();
}
int tm_export_game(const tm_store* s, int game, int32_t* child, float* score, int32_t* n_to_o, int32_t* visit,
                   float* value, float* variance, uint8_t* end_obs, void* stream) {
    hipLaunchKernelGGL(k_export_game, dim3((s->max_nodes + 255) / 256), dim3(256), 0, (hipStream_t)stream, *s, game,
                       child, score, n_to_o, visit, value, variance, end_obs);
    return TM_LAUNCH_CHECK();
}

}  // extern "C"

// layout self-description so the host mirror (ctypes) can be checked without a GPU
extern "C" int tm_store_layout(int* out, int n) {
    int v[] = {(int)sizeof(tm_store), (int)offsetof(tm_store, gamma), (int)offsetof(tm_store, node_rec),
               (int)offsetof(tm_store, nq_table), (int)offsetof(tm_store, replay_count), (int)offsetof(tm_store, mt_state),
               (int)offsetof(tm_store, node_child), (int)offsetof(tm_store, gc_slice_cycles),
               (int)offsetof(tm_store, gc_part), (int)offsetof(tm_store, dist_vmin)};
    int m = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; i < n && i < m; ++i) out[i] = v[i];
    return m;
}
