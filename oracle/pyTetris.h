/*
 * ORACLE — test infrastructure only (see oracle/tetris_engine.h).
 *
 * C++ `Tetris` type with exactly the surface the reference's native agent needs from the absent
 * pyTetris package (discovered from agents/cppmodule/agent.cpp:32-39,94-95,201-264,420,440 and
 * agents/cppmodule/core.h:12): `end`, `score`, default ctor, `hash()`, `==`, `copy_from`,
 * `play`, `_getState()` (200 chars), `getState()` (20x10 numpy, resizable).  Behaviour follows
 * ENGINE_SPEC.md via oracle/tetris_engine.h.  Naming this file pyTetris.h lets the UNMODIFIED
 * reference sources core.cpp / agent.cpp compile against it (oracle/Makefile -> oracle/_ref/).
 */
#ifndef PYTETRIS_H
#define PYTETRIS_H
#include <cstdint>
#include <cstdio>
#include <vector>
#include <pybind11/pybind11.h>
#include <pybind11/numpy.h>
#include "tetris_engine.h"

struct Tetris {
    ot_game g;
    ot_cfg cfg;
    int32_t line_stats[4];
    /* mirrors the reference code reads as plain members */
    bool end;
    int score;
    int line_clears;
    int combo;

    Tetris() : Tetris(1, 0, 0, 0) {}
    Tetris(int app, int scoring, int randomizer, uint32_t seed) {
        cfg.app = app; cfg.scoring = scoring; cfg.randomizer = randomizer;
        for (int i = 0; i < 4; ++i) line_stats[i] = 0;
        ot_init(&g, &cfg, seed);
        sync();
    }
    void sync() {
        end = (g.flags & 1) != 0;
        score = g.score;
        line_clears = g.line_clears;
        combo = g.combo;
    }
    size_t hash() const { return (size_t)ot_hash_game(&g); }
    bool operator==(const Tetris &o) const { return std::memcmp(&g, &o.g, sizeof(g)) == 0; }
    void copy_from(const Tetris &o) {
        g = o.g; cfg = o.cfg;
        for (int i = 0; i < 4; ++i) line_stats[i] = o.line_stats[i];
        sync();
    }
    void play(int a) { ot_play(&g, &cfg, a, line_stats); sync(); }
    void reset() { ot_reset(&g, &cfg, line_stats); sync(); }
    void seed(uint32_t s) {
        for (int i = 0; i < 4; ++i) line_stats[i] = 0;
        ot_init(&g, &cfg, s);
        sync();
    }
    std::vector<char> _getState() const {
        std::vector<char> v(200);
        ot_render(&g, reinterpret_cast<int8_t *>(v.data()));
        return v;
    }
    pybind11::array_t<int8_t> getState() const {
        pybind11::array_t<int8_t> a({20, 10});
        ot_render(&g, a.mutable_data());
        return a;
    }
    void printState() const {
        int8_t s[200];
        ot_render(&g, s);
        for (int r = 0; r < 20; ++r) {
            for (int c = 0; c < 10; ++c) std::putchar(s[r * 10 + c] == 0 ? '.' : (s[r * 10 + c] == 1 ? '#' : '@'));
            std::putchar('\n');
        }
        std::fflush(stdout);
    }
};
#endif
