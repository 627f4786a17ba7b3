/*
 * ORACLE — test infrastructure only.  Nothing under tetris_mcts_amd/ may include, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * CPU restatement of the Tetris environment the reference calls through the external
 * `pyTetris` package (reference call sites: play.py:75-76,150,155-177;
 * agents/agent.py:101-129,140-145; agents/cppmodule/agent.cpp:94-95,201-264).
 * pyTetris==1.0.0 (requirements.txt:15) is absent from /root/reference and nothing in the
 * reference pins its results, so this engine's parity with pyTetris is UNPINNED: it follows
 * ENGINE_SPEC.md (this repo's own specification) section by section.  It is written cell by
 * cell on purpose - an independent formulation from the bit-mask engine in
 * tetris_mcts_amd/csrc/engine.h that it checks.
 */
#ifndef ORACLE_TETRIS_ENGINE_H
#define ORACLE_TETRIS_ENGINE_H
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OT_ROWS 20
#define OT_COLS 10

/* ENGINE_SPEC.md section 2: the 64-byte packed game. */
typedef struct {
    uint16_t rows[OT_ROWS];
    uint8_t piece, rot;
    int8_t x, y;
    uint8_t drop_ctr, flags; /* bit0 end, bit1 back-to-back */
    int16_t combo;
    uint32_t piece_count, seed;
    int32_t score, line_clears;
} ot_game;

typedef struct {
    int app, scoring, randomizer;
} ot_cfg;

/* ENGINE_SPEC.md section 3: (col,row) of the 4 cells, [piece][rot][cell][2]. */
static const int8_t OT_CELLS[7][4][4][2] = {
    /* I */ {{{0,1},{1,1},{2,1},{3,1}}, {{2,0},{2,1},{2,2},{2,3}}, {{0,2},{1,2},{2,2},{3,2}}, {{1,0},{1,1},{1,2},{1,3}}},
    /* O */ {{{1,0},{2,0},{1,1},{2,1}}, {{1,0},{2,0},{1,1},{2,1}}, {{1,0},{2,0},{1,1},{2,1}}, {{1,0},{2,0},{1,1},{2,1}}},
    /* T */ {{{1,0},{0,1},{1,1},{2,1}}, {{1,0},{1,1},{2,1},{1,2}}, {{0,1},{1,1},{2,1},{1,2}}, {{1,0},{0,1},{1,1},{1,2}}},
    /* S */ {{{1,0},{2,0},{0,1},{1,1}}, {{1,0},{1,1},{2,1},{2,2}}, {{1,1},{2,1},{0,2},{1,2}}, {{0,0},{0,1},{1,1},{1,2}}},
    /* Z */ {{{0,0},{1,0},{1,1},{2,1}}, {{2,0},{1,1},{2,1},{1,2}}, {{0,1},{1,1},{1,2},{2,2}}, {{1,0},{0,1},{1,1},{0,2}}},
    /* J */ {{{0,0},{0,1},{1,1},{2,1}}, {{1,0},{2,0},{1,1},{1,2}}, {{0,1},{1,1},{2,1},{2,2}}, {{1,0},{1,1},{0,2},{1,2}}},
    /* L */ {{{2,0},{0,1},{1,1},{2,1}}, {{1,0},{1,1},{1,2},{2,2}}, {{0,1},{1,1},{2,1},{0,2}}, {{0,0},{1,0},{1,1},{1,2}}},
};

/* ENGINE_SPEC.md section 4: SRS kicks (dx, dy-up), index [from_rot*2 + (ccw?1:0)][test]. */
static const int8_t OT_KICK_JLSTZ[8][5][2] = {
    /* 0>1 */ {{0,0},{-1,0},{-1,1},{0,-2},{-1,-2}}, /* 0>3 */ {{0,0},{1,0},{1,1},{0,-2},{1,-2}},
    /* 1>2 */ {{0,0},{1,0},{1,-1},{0,2},{1,2}},     /* 1>0 */ {{0,0},{1,0},{1,-1},{0,2},{1,2}},
    /* 2>3 */ {{0,0},{1,0},{1,1},{0,-2},{1,-2}},    /* 2>1 */ {{0,0},{-1,0},{-1,1},{0,-2},{-1,-2}},
    /* 3>0 */ {{0,0},{-1,0},{-1,-1},{0,2},{-1,2}},  /* 3>2 */ {{0,0},{-1,0},{-1,-1},{0,2},{-1,2}},
};
static const int8_t OT_KICK_I[8][5][2] = {
    /* 0>1 */ {{0,0},{-2,0},{1,0},{-2,-1},{1,2}},   /* 0>3 */ {{0,0},{-1,0},{2,0},{-1,2},{2,-1}},
    /* 1>2 */ {{0,0},{-1,0},{2,0},{-1,2},{2,-1}},   /* 1>0 */ {{0,0},{2,0},{-1,0},{2,1},{-1,-2}},
    /* 2>3 */ {{0,0},{2,0},{-1,0},{2,1},{-1,-2}},   /* 2>1 */ {{0,0},{1,0},{-2,0},{1,-2},{-2,1}},
    /* 3>0 */ {{0,0},{1,0},{-2,0},{1,-2},{-2,1}},   /* 3>2 */ {{0,0},{-2,0},{1,0},{-2,-1},{1,2}},
};

static inline uint64_t ot_splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint32_t ot_rnd(uint32_t seed, uint32_t ctr) {
    return (uint32_t)(ot_splitmix64(((uint64_t)seed << 32) | ctr) >> 32);
}

/* ENGINE_SPEC.md section 6. */
static inline int ot_piece_at(uint32_t seed, uint32_t i, int randomizer) {
    if (randomizer == 1) return (int)(ot_rnd(seed, i) % 7u);
    uint32_t k = i / 7u;
    int perm[7] = {0, 1, 2, 3, 4, 5, 6};
    for (int t = 6; t >= 1; --t) {
        uint32_t r = ot_rnd(seed, 8u * k + (uint32_t)t) % (uint32_t)(t + 1);
        int tmp = perm[t]; perm[t] = perm[r]; perm[r] = tmp;
    }
    return perm[i % 7u];
}

static inline int ot_cell_blocked(const ot_game *g, int row, int col) {
    if (col < 0 || col >= OT_COLS || row < 0 || row >= OT_ROWS) return 1;
    return (g->rows[row] >> col) & 1;
}
static inline int ot_collides(const ot_game *g, int piece, int rot, int x, int y) {
    for (int c = 0; c < 4; ++c)
        if (ot_cell_blocked(g, y + OT_CELLS[piece][rot][c][1], x + OT_CELLS[piece][rot][c][0])) return 1;
    return 0;
}

static inline void ot_spawn(ot_game *g, const ot_cfg *cfg) {
    g->piece = (uint8_t)ot_piece_at(g->seed, g->piece_count, cfg->randomizer);
    g->piece_count += 1;
    g->rot = 0; g->x = 3; g->y = 0; g->drop_ctr = 0;
    if (ot_collides(g, g->piece, 0, 3, 0)) g->flags |= 1;
}

static inline void ot_init(ot_game *g, const ot_cfg *cfg, uint32_t seed) {
    memset(g, 0, sizeof(*g));
    g->seed = seed;
    g->combo = -1;
    ot_spawn(g, cfg);
}
/* line_stats may be NULL */
static inline void ot_reset(ot_game *g, const ot_cfg *cfg, int32_t *line_stats) {
    uint32_t s = ot_rnd(g->seed, 0xFFFFFFFFu);
    ot_init(g, cfg, s);
    if (line_stats) memset(line_stats, 0, 4 * sizeof(int32_t));
}

/* ENGINE_SPEC.md section 5. */
static inline void ot_lock(ot_game *g, const ot_cfg *cfg, int32_t *line_stats) {
    for (int c = 0; c < 4; ++c) {
        int row = g->y + OT_CELLS[g->piece][g->rot][c][1];
        int col = g->x + OT_CELLS[g->piece][g->rot][c][0];
        g->rows[row] |= (uint16_t)(1u << col);
    }
    int n = 0;
    int dst = OT_ROWS - 1;
    for (int r = OT_ROWS - 1; r >= 0; --r) {
        if (g->rows[r] == 0x3FF) { n += 1; continue; }
        g->rows[dst--] = g->rows[r];
    }
    while (dst >= 0) g->rows[dst--] = 0;
    g->line_clears += n;
    if (n > 0) {
        if (line_stats) line_stats[n - 1] += 1;
        g->combo += 1;
        if (cfg->scoring == 1) {
            g->score += n;
        } else {
            static const int base[5] = {0, 100, 300, 500, 800};
            int b = base[n];
            if (n == 4 && (g->flags & 2)) b = 1200;
            g->score += b + 50 * g->combo;
            if (n == 4) g->flags |= 2; else g->flags &= (uint8_t)~2;
        }
    } else {
        g->combo = -1;
    }
    ot_spawn(g, cfg);
}

/* ENGINE_SPEC.md section 4. */
static inline void ot_play(ot_game *g, const ot_cfg *cfg, int a, int32_t *line_stats) {
    if (g->flags & 1) return;
    int locked = 0;
    switch (a) {
    case 1: if (!ot_collides(g, g->piece, g->rot, g->x - 1, g->y)) g->x -= 1; break;
    case 2: if (!ot_collides(g, g->piece, g->rot, g->x + 1, g->y)) g->x += 1; break;
    case 3:
        while (!ot_collides(g, g->piece, g->rot, g->x, g->y + 1)) {
            g->y += 1;
            if (cfg->scoring == 0) g->score += 2;
        }
        ot_lock(g, cfg, line_stats);
        locked = 1;
        break;
    case 4:
        if (!ot_collides(g, g->piece, g->rot, g->x, g->y + 1)) {
            g->y += 1;
            if (cfg->scoring == 0) g->score += 1;
        } else {
            ot_lock(g, cfg, line_stats);
            locked = 1;
        }
        break;
    case 5: case 6:
        if (g->piece != 1) {
            int ccw = (a == 6);
            int nr = (g->rot + (ccw ? 3 : 1)) & 3;
            const int8_t (*k)[2] = (g->piece == 0 ? OT_KICK_I : OT_KICK_JLSTZ)[g->rot * 2 + ccw];
            for (int t = 0; t < 5; ++t) {
                if (k[t][1] > 0) continue; /* upward kicks are not allowed: keeps the state graph acyclic */
                int nx = g->x + k[t][0], ny = g->y - k[t][1];
                if (!ot_collides(g, g->piece, nr, nx, ny)) {
                    g->rot = (uint8_t)nr; g->x = (int8_t)nx; g->y = (int8_t)ny;
                    break;
                }
            }
        }
        break;
    default: break;
    }
    if (locked) return;
    g->drop_ctr += 1;
    if (g->drop_ctr >= cfg->app) {
        g->drop_ctr = 0;
        if (!ot_collides(g, g->piece, g->rot, g->x, g->y + 1)) g->y += 1;
        else ot_lock(g, cfg, line_stats);
    }
}

/* ENGINE_SPEC.md section 7. */
static inline void ot_render(const ot_game *g, int8_t *out /* [20*10] */) {
    for (int r = 0; r < OT_ROWS; ++r)
        for (int c = 0; c < OT_COLS; ++c) out[r * OT_COLS + c] = (int8_t)((g->rows[r] >> c) & 1);
    if (!(g->flags & 1))
        for (int c = 0; c < 4; ++c) {
            int row = g->y + OT_CELLS[g->piece][g->rot][c][1];
            int col = g->x + OT_CELLS[g->piece][g->rot][c][0];
            out[row * OT_COLS + col] = -1;
        }
}

typedef struct {
    uint16_t rows[OT_ROWS];
    uint8_t cells[4];
    uint8_t end, pad[3];
} ot_obs;

static inline void ot_pack_obs(const ot_game *g, ot_obs *o) {
    memset(o, 0, sizeof(*o));
    memcpy(o->rows, g->rows, sizeof(o->rows));
    if (g->flags & 1) {
        o->cells[0] = o->cells[1] = o->cells[2] = o->cells[3] = 0xFF;
        o->end = 1;
        return;
    }
    for (int c = 0; c < 4; ++c) {
        int row = g->y + OT_CELLS[g->piece][g->rot][c][1];
        int col = g->x + OT_CELLS[g->piece][g->rot][c][0];
        o->cells[c] = (uint8_t)(row * OT_COLS + col);
    }
    for (int i = 1; i < 4; ++i) /* insertion sort ascending */
        for (int j = i; j > 0 && o->cells[j - 1] > o->cells[j]; --j) {
            uint8_t t = o->cells[j]; o->cells[j] = o->cells[j - 1]; o->cells[j - 1] = t;
        }
}
static inline void ot_render_obs(const ot_obs *o, int8_t *out) {
    for (int r = 0; r < OT_ROWS; ++r)
        for (int c = 0; c < OT_COLS; ++c) out[r * OT_COLS + c] = (int8_t)((o->rows[r] >> c) & 1);
    if (!o->end)
        for (int c = 0; c < 4; ++c) out[o->cells[c]] = -1;
}

/* ENGINE_SPEC.md section 8. */
static inline uint64_t ot_hash_words(const void *p, int nwords, uint64_t h) {
    const unsigned char *b = (const unsigned char *)p;
    for (int i = 0; i < nwords; ++i) {
        uint64_t w;
        memcpy(&w, b + 8 * i, 8);
        h = ot_splitmix64(h ^ w);
    }
    return h;
}
static inline uint64_t ot_hash_game(const ot_game *g) { return ot_hash_words(g, 8, 0x243F6A8885A308D3ULL); }
static inline uint64_t ot_hash_obs(const ot_obs *o) { return ot_hash_words(o, 6, 0x13198A2E03707344ULL); }

#ifdef __cplusplus
}
#endif
#endif
