// Tetris board step for gfx950 (ENGINE_SPEC.md), the "T" row of SURVEY.md section 8a: what the
// reference reaches through the external pyTetris package (call sites agents/agent.py:101-129,140-145,
// agents/cppmodule/agent.cpp:201-264).
//
// Formulation: a game is the 64-byte packed record of ENGINE_SPEC.md section 2, resident in LDS while
// it is stepped (one 64-byte slot per lane that steps a game).  Rows are 10-bit masks; a piece
// orientation is a 16-bit 4x4 mask; collision is an AND of the piece's four box rows against
// wall-padded board rows (3 wall bits each side, solid sentinel rows above and below), so wall,
// floor, ceiling and cell tests are the same bit test.  Lock ORs the box rows in, line clear is a
// bottom-up compaction of the 20 row words.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmcts {

// ---- packed game (ENGINE_SPEC.md section 2), as 16 dwords ----
// dword 0..9  : rows[20] (u16 pairs)
// dword 10    : piece | rot<<8 | (x&0xFF)<<16 | (y&0xFF)<<24
// dword 11    : drop_ctr | flags<<8 | (combo&0xFFFF)<<16
// dword 12    : piece_count   13: seed   14: score   15: line_clears
constexpr int GAME_DW = 16;
constexpr int OBS_DW = 12;   // packed observation (ENGINE_SPEC.md section 7): rows[20], cells[4], end, pad[3]

struct EngCfg {
    int app, scoring, randomizer;
};

// piece orientation masks: bit (4*row + col) of the 4x4 box, built from the cell lists of
// ENGINE_SPEC.md section 3 at compile time.
struct CellList { int8_t c[4][2]; };
constexpr uint16_t mk(const CellList& l) {
    uint16_t m = 0;
    for (int i = 0; i < 4; ++i) m = (uint16_t)(m | (1u << (4 * l.c[i][1] + l.c[i][0])));
    return m;
}
#define TM_CL(a, b, c, d, e, f, g, h) mk(CellList{{{a, b}, {c, d}, {e, f}, {g, h}}})
constexpr uint16_t PIECE_MASK_C[7][4] = {
    /* I */ {TM_CL(0,1,1,1,2,1,3,1), TM_CL(2,0,2,1,2,2,2,3), TM_CL(0,2,1,2,2,2,3,2), TM_CL(1,0,1,1,1,2,1,3)},
    /* O */ {TM_CL(1,0,2,0,1,1,2,1), TM_CL(1,0,2,0,1,1,2,1), TM_CL(1,0,2,0,1,1,2,1), TM_CL(1,0,2,0,1,1,2,1)},
    /* T */ {TM_CL(1,0,0,1,1,1,2,1), TM_CL(1,0,1,1,2,1,1,2), TM_CL(0,1,1,1,2,1,1,2), TM_CL(1,0,0,1,1,1,1,2)},
    /* S */ {TM_CL(1,0,2,0,0,1,1,1), TM_CL(1,0,1,1,2,1,2,2), TM_CL(1,1,2,1,0,2,1,2), TM_CL(0,0,0,1,1,1,1,2)},
    /* Z */ {TM_CL(0,0,1,0,1,1,2,1), TM_CL(2,0,1,1,2,1,1,2), TM_CL(0,1,1,1,1,2,2,2), TM_CL(1,0,0,1,1,1,0,2)},
    /* J */ {TM_CL(0,0,0,1,1,1,2,1), TM_CL(1,0,2,0,1,1,1,2), TM_CL(0,1,1,1,2,1,2,2), TM_CL(1,0,1,1,0,2,1,2)},
    /* L */ {TM_CL(2,0,0,1,1,1,2,1), TM_CL(1,0,1,1,1,2,2,2), TM_CL(0,1,1,1,2,1,0,2), TM_CL(0,0,1,0,1,1,1,2)},
};
#undef TM_CL
// the four cells of an orientation as box positions (4*row + col), ascending, one nibble each: what pack_obs walks
constexpr uint16_t mk_cells(uint16_t mask) {
    uint16_t out = 0;
    int n = 0;
    for (int bit = 0; bit < 16; ++bit)
        if ((mask >> bit) & 1u) { out = (uint16_t)(out | (bit << (4 * n))); n += 1; }
    return out;
}
#define TM_ROW(f, p) {f(PIECE_MASK_C[p][0]), f(PIECE_MASK_C[p][1]), f(PIECE_MASK_C[p][2]), f(PIECE_MASK_C[p][3])}
#define TM_ID(x) (x)
__device__ const uint16_t PIECE_MASK[7][4] = {TM_ROW(TM_ID, 0), TM_ROW(TM_ID, 1), TM_ROW(TM_ID, 2), TM_ROW(TM_ID, 3),
                                              TM_ROW(TM_ID, 4), TM_ROW(TM_ID, 5), TM_ROW(TM_ID, 6)};
__device__ const uint16_t PIECE_CELLS[7][4] = {TM_ROW(mk_cells, 0), TM_ROW(mk_cells, 1), TM_ROW(mk_cells, 2), TM_ROW(mk_cells, 3),
                                               TM_ROW(mk_cells, 4), TM_ROW(mk_cells, 5), TM_ROW(mk_cells, 6)};
#undef TM_ROW
#undef TM_ID

// SRS kicks (dx, dy-up), index [set][from_rot*2 + ccw][test]; set 0 = JLSTZ, 1 = I (ENGINE_SPEC.md section 4)
__device__ const int8_t KICKS[2][8][5][2] = {
    {{{0,0},{-1,0},{-1,1},{0,-2},{-1,-2}}, {{0,0},{1,0},{1,1},{0,-2},{1,-2}},
     {{0,0},{1,0},{1,-1},{0,2},{1,2}},     {{0,0},{1,0},{1,-1},{0,2},{1,2}},
     {{0,0},{1,0},{1,1},{0,-2},{1,-2}},    {{0,0},{-1,0},{-1,1},{0,-2},{-1,-2}},
     {{0,0},{-1,0},{-1,-1},{0,2},{-1,2}},  {{0,0},{-1,0},{-1,-1},{0,2},{-1,2}}},
    {{{0,0},{-2,0},{1,0},{-2,-1},{1,2}},   {{0,0},{-1,0},{2,0},{-1,2},{2,-1}},
     {{0,0},{-1,0},{2,0},{-1,2},{2,-1}},   {{0,0},{2,0},{-1,0},{2,1},{-1,-2}},
     {{0,0},{2,0},{-1,0},{2,1},{-1,-2}},   {{0,0},{1,0},{-2,0},{1,-2},{-2,1}},
     {{0,0},{1,0},{-2,0},{1,-2},{-2,1}},   {{0,0},{-2,0},{1,0},{-2,-1},{1,2}}},
};

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t rnd32(uint32_t seed, uint32_t ctr) {
    return (uint32_t)(splitmix64(((uint64_t)seed << 32) | ctr) >> 32);
}
// ENGINE_SPEC.md section 6
__device__ __forceinline__ int piece_at(uint32_t seed, uint32_t i, int randomizer) {
    if (randomizer == 1) return (int)(rnd32(seed, i) % 7u);
    uint32_t k = i / 7u;
    // perm packed as 7 nibbles
    uint32_t perm = 0x6543210u;
    for (int t = 6; t >= 1; --t) {
        uint32_t r = rnd32(seed, 8u * k + (uint32_t)t) % (uint32_t)(t + 1);
        uint32_t a = (perm >> (4 * t)) & 0xFu, b = (perm >> (4 * r)) & 0xFu;
        perm &= ~((0xFu << (4 * t)) | (0xFu << (4 * r)));
        perm |= b << (4 * t);
        if ((uint32_t)t != r) perm |= a << (4 * r);
    }
    return (int)((perm >> (4 * (i % 7u))) & 0xFu);
}

// A game being stepped: rows live in LDS (u16[20] at the start of the lane's 64-byte slot),
// the scalar fields live in registers and are written back by store_fields().
struct Piece {
    int piece, rot, x, y;
    int drop_ctr, flags, combo;
    uint32_t piece_count, seed;
    int score, line_clears;
};

__device__ __forceinline__ void load_fields(const uint32_t* slot, Piece& p) {
    uint32_t a = slot[10], b = slot[11];
    p.piece = a & 0xFF; p.rot = (a >> 8) & 0xFF;
    p.x = (int)(int8_t)((a >> 16) & 0xFF); p.y = (int)(int8_t)((a >> 24) & 0xFF);
    p.drop_ctr = b & 0xFF; p.flags = (b >> 8) & 0xFF; p.combo = (int)(int16_t)(b >> 16);
    p.piece_count = slot[12]; p.seed = slot[13]; p.score = (int)slot[14]; p.line_clears = (int)slot[15];
}
__device__ __forceinline__ void store_fields(uint32_t* slot, const Piece& p) {
    slot[10] = (uint32_t)(p.piece & 0xFF) | ((uint32_t)(p.rot & 0xFF) << 8) | ((uint32_t)(p.x & 0xFF) << 16) |
               ((uint32_t)(p.y & 0xFF) << 24);
    slot[11] = (uint32_t)(p.drop_ctr & 0xFF) | ((uint32_t)(p.flags & 0xFF) << 8) | ((uint32_t)(p.combo & 0xFFFF) << 16);
    slot[12] = p.piece_count; slot[13] = p.seed; slot[14] = (uint32_t)p.score; slot[15] = (uint32_t)p.line_clears;
}

// wall-padded row: bits 0..2 and 13..15 are walls, column c sits at bit c+3; rows outside 0..19 are solid
__device__ __forceinline__ uint32_t prow(const uint16_t* rows, int r) {
    if (r < 0 || r > 19) return 0xFFFFu;
    return ((uint32_t)rows[r] << 3) | 0xE007u;
}
__device__ __forceinline__ bool collides(const uint16_t* rows, uint32_t mask, int x, int y) {
    uint32_t hit = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t bits = (mask >> (4 * i)) & 0xFu;
        hit |= (prow(rows, y + i) >> (x + 3)) & bits;
    }
    return hit != 0;
}

// piece_at(seed, i) by the lanes of a wave together (every lane calls it with the same wave-uniform arguments and gets the
// result): the six draws of the 7-bag shuffle are independent - lane t computes r_t, the six swaps follow on wave-uniform
// values.  The seven successors of an expansion share seed and piece count, and at least one of them (the hard drop) locks
// and spawns in every expansion: one such call replaces a 300-instruction shuffle on a single lane.
__device__ __forceinline__ int wave_piece_at(uint32_t seed, uint32_t i, int randomizer, int lane) {
    if (randomizer == 1) return (int)(rnd32(seed, i) % 7u);
    const uint32_t k = i / 7u;
    const uint32_t t_l = (uint32_t)(lane >= 1 && lane <= 6 ? lane : 1);
    const uint32_t r_l = rnd32(seed, 8u * k + t_l) % (t_l + 1u);
    uint32_t perm = 0x6543210u;
    for (int t = 6; t >= 1; --t) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)r_l, t);
        const uint32_t a = (perm >> (4 * t)) & 0xFu, b = (perm >> (4 * r)) & 0xFu;
        perm &= ~((0xFu << (4 * t)) | (0xFu << (4 * r)));
        perm |= b << (4 * t);
        if ((uint32_t)t != r) perm |= a << (4 * r);
    }
    return (int)((perm >> (4 * (i % 7u))) & 0xFu);
}

// `next_piece` >= 0: piece_at(p.seed, p.piece_count) as the caller has already computed it (wave_piece_at)
__device__ __forceinline__ void spawn(const uint16_t* rows, Piece& p, const EngCfg& cfg, int next_piece = -1) {
    p.piece = next_piece >= 0 ? next_piece : piece_at(p.seed, p.piece_count, cfg.randomizer);
    p.piece_count += 1;
    p.rot = 0; p.x = 3; p.y = 0; p.drop_ctr = 0;
    if (collides(rows, PIECE_MASK[p.piece][0], 3, 0)) p.flags |= 1;
}

// ENGINE_SPEC.md section 5.  line_stats (may be null) is a per-environment side record.
__device__ __forceinline__ void lock_piece(uint16_t* rows, Piece& p, const EngCfg& cfg, int* line_stats, int next_piece = -1) {
    uint32_t mask = PIECE_MASK[p.piece][p.rot];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t bits = (mask >> (4 * i)) & 0xFu;
        if (bits) {
            int r = p.y + i;   // always inside 0..19 for a non-colliding piece
            rows[r] = (uint16_t)(rows[r] | (((bits << (p.x + 3)) >> 3) & 0x3FFu));
        }
    }
    // only the (at most four) rows the piece touched can have become full: most locks clear nothing and skip the compaction
    int n = 0;
    bool any_full = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = p.y + i;
        if (((mask >> (4 * i)) & 0xFu) && rows[r] == 0x3FF) any_full = true;
    }
    if (any_full) {
        int dst = 19;
        for (int r = 19; r >= 0; --r) {
            uint16_t v = rows[r];
            if (v == 0x3FF) { n += 1; continue; }
            rows[dst--] = v;
        }
        for (; dst >= 0; --dst) rows[dst] = 0;
    }
    p.line_clears += n;
    if (n > 0) {
        if (line_stats) line_stats[n - 1] += 1;
        p.combo += 1;
        if (cfg.scoring == 1) {
            p.score += n;
        } else {
            int b = (n == 1) ? 100 : (n == 2) ? 300 : (n == 3) ? 500 : 800;
            if (n == 4 && (p.flags & 2)) b = 1200;
            p.score += b + 50 * p.combo;
            if (n == 4) p.flags |= 2; else p.flags &= ~2;
        }
    } else {
        p.combo = -1;
    }
    spawn(rows, p, cfg, next_piece);
}

// ENGINE_SPEC.md section 4: one play(a).  The piece locks at ONE place in the code: the three ways to lock (hard drop, a
// blocked soft drop, gravity) only raise a flag, so lanes of a wavefront that step different games with different actions
// run lock + line clear + spawn (the expensive part: the 7-bag shuffle is six 64-bit hashes) together, once, instead of
// once per call site.
// `drop_rows` >= 0: the caller already knows how many rows the piece can fall (computed by the lanes of a wave together,
// hard_drop_rows below); -1: found here, row by row.
// `next_piece` >= 0: the piece a lock would spawn, piece_at(p.seed, p.piece_count), already computed (wave_piece_at).
__device__ __forceinline__ void play(uint16_t* rows, Piece& p, const EngCfg& cfg, int a, int* line_stats, int drop_rows = -1,
                                     int next_piece = -1) {
    if (p.flags & 1) return;
    bool do_lock = false;
    uint32_t mask = PIECE_MASK[p.piece][p.rot];
    if (a == 1) {
        if (!collides(rows, mask, p.x - 1, p.y)) p.x -= 1;
    } else if (a == 2) {
        if (!collides(rows, mask, p.x + 1, p.y)) p.x += 1;
    } else if (a == 3) {
        if (drop_rows >= 0) {
            p.y += drop_rows;
            if (cfg.scoring == 0) p.score += 2 * drop_rows;
        } else {
            while (!collides(rows, mask, p.x, p.y + 1)) {
                p.y += 1;
                if (cfg.scoring == 0) p.score += 2;
            }
        }
        do_lock = true;
    } else if (a == 4) {
        if (!collides(rows, mask, p.x, p.y + 1)) {
            p.y += 1;
            if (cfg.scoring == 0) p.score += 1;
        } else {
            do_lock = true;
        }
    } else if (a == 5 || a == 6) {
        if (p.piece != 1) {
            int ccw = (a == 6);
            int nr = (p.rot + (ccw ? 3 : 1)) & 3;
            uint32_t nmask = PIECE_MASK[p.piece][nr];
            const int8_t(*k)[2] = KICKS[p.piece == 0 ? 1 : 0][p.rot * 2 + ccw];
            for (int t = 0; t < 5; ++t) {
                int dx = k[t][0], dy = k[t][1];
                if (dy > 0) continue;   // no upward kicks: the state graph stays a DAG
                int nx = p.x + dx, ny = p.y - dy;
                if (!collides(rows, nmask, nx, ny)) { p.rot = nr; p.x = nx; p.y = ny; break; }
            }
        }
    }
    if (!do_lock) {      // an action that locked the piece is not followed by gravity
        p.drop_ctr += 1;
        if (p.drop_ctr >= cfg.app) {
            p.drop_ctr = 0;
            if (!collides(rows, PIECE_MASK[p.piece][p.rot], p.x, p.y + 1)) p.y += 1;
            else do_lock = true;
        }
    }
    if (do_lock) lock_piece(rows, p, cfg, line_stats, next_piece);
}

// How many rows the falling piece of the game in `slot` can fall (the hard drop's loop), by the lanes of a wave together:
// lane r tests the piece r+1 rows down (rows beyond the floor are solid), the first lane that collides gives the answer.
__device__ __forceinline__ int hard_drop_rows(const uint32_t* slot, int lane) {
    Piece p;
    load_fields(slot, p);
    const bool hit = lane > 20 || collides(reinterpret_cast<const uint16_t*>(slot), PIECE_MASK[p.piece][p.rot], p.x, p.y + 1 + lane);
    return __ffsll((long long)__ballot(hit)) - 1;
}

// new environment (ENGINE_SPEC.md section 6: spawns piece(0))
__device__ inline void init_game(uint32_t* slot, const EngCfg& cfg, uint32_t seed) {
    for (int i = 0; i < GAME_DW; ++i) slot[i] = 0;
    Piece p{};
    p.seed = seed; p.combo = -1;
    spawn(reinterpret_cast<const uint16_t*>(slot), p, cfg);
    store_fields(slot, p);
}

// ENGINE_SPEC.md section 8
__device__ __forceinline__ uint64_t hash_game(const uint32_t* s) {
    uint64_t h = 0x243F6A8885A308D3ULL;
#pragma unroll
    for (int i = 0; i < 8; ++i) h = splitmix64(h ^ (((uint64_t)s[2 * i + 1] << 32) | s[2 * i]));
    return h;
}
__device__ __forceinline__ uint64_t hash_obs(const uint32_t* s) {
    uint64_t h = 0x13198A2E03707344ULL;
#pragma unroll
    for (int i = 0; i < 6; ++i) h = splitmix64(h ^ (((uint64_t)s[2 * i + 1] << 32) | s[2 * i]));
    return h;
}

// ENGINE_SPEC.md section 7: packed observation from a packed game (both as dword arrays)
__device__ inline void pack_obs(const uint32_t* g, uint32_t* o) {
#pragma unroll
    for (int i = 0; i < 10; ++i) o[i] = g[i];
    uint32_t a = g[10], b = g[11];
    if ((b >> 8) & 1) { o[10] = 0xFFFFFFFFu; o[11] = 1u; return; }
    int piece = a & 0xFF, rot = (a >> 8) & 0xFF;
    int x = (int)(int8_t)((a >> 16) & 0xFF), y = (int)(int8_t)((a >> 24) & 0xFF);
    // the piece's four cells as row * 10 + col, ascending (box positions ascend row-major, which is ascending row * 10 + col)
    const uint32_t pos = PIECE_CELLS[piece][rot];
    uint32_t cells = 0;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int bit = (int)((pos >> (4 * n)) & 0xFu);
        cells |= (uint32_t)((y + (bit >> 2)) * 10 + x + (bit & 3)) << (8 * n);
    }
    o[10] = cells;
    o[11] = 0;
}

// render a packed observation as int8[200] values for cell index i (0 empty, 1 locked, -1 falling)
__device__ __forceinline__ int obs_cell(const uint32_t* o, int i) {
    int r = i / 10, c = i - 10 * r;
    uint32_t w = o[r >> 1];
    int v = (int)((w >> (16 * (r & 1) + c)) & 1u);
    if (!(o[11] & 0xFFu)) {
        uint32_t cells = o[10];
        if (((cells & 0xFF) == (uint32_t)i) | (((cells >> 8) & 0xFF) == (uint32_t)i) |
            (((cells >> 16) & 0xFF) == (uint32_t)i) | ((cells >> 24) == (uint32_t)i))
            v = -1;
    }
    return v;
}

}  // namespace tmcts
