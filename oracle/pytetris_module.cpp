/*
 * ORACLE — test infrastructure only (see oracle/tetris_engine.h).
 * pybind11 module named `pyTetris` so the reference's Python code (`from pyTetris import Tetris`,
 * play.py:1) and its agents (agents/agent.py:70,101-129) run unmodified on the oracle engine.
 * The buffer protocol exposes the C++ object itself because the reference's native agent
 * reinterprets `buffer.ptr` as `Tetris*` (agents/cppmodule/agent.cpp:210-214,266-282).
 */
#include <pybind11/pybind11.h>
#include <pybind11/numpy.h>
#include <pybind11/stl.h>
#include "pyTetris.h"

namespace py = pybind11;

PYBIND11_MODULE(pyTetris, m) {
    py::class_<Tetris>(m, "Tetris", py::buffer_protocol())
        .def(py::init([](py::object shape, int app, int scoring, int randomizer, uint32_t seed) {
                 if (!shape.is_none()) {
                     auto t = shape.cast<std::pair<int, int>>();
                     if (t.first != 20 || t.second != 10) throw std::invalid_argument("shape must be (20, 10)");
                 }
                 return new Tetris(app, scoring, randomizer, seed);
             }),
             py::arg("shape") = py::none(), py::arg("actions_per_drop") = 1, py::arg("scoring") = 0,
             py::arg("randomizer") = 0, py::arg("seed") = 0)
        .def_buffer([](Tetris &t) {
            return py::buffer_info(reinterpret_cast<void *>(&t), 1, py::format_descriptor<uint8_t>::format(), 1,
                                   {sizeof(Tetris)}, {1});
        })
        .def("play", &Tetris::play)
        .def("copy_from", &Tetris::copy_from)
        .def("reset", &Tetris::reset)
        .def("seed", &Tetris::seed)
        .def("getState", &Tetris::getState)
        .def("printState", &Tetris::printState)
        .def("packed", [](const Tetris &t) { return py::bytes(reinterpret_cast<const char *>(&t.g), sizeof(t.g)); })
        .def("packed_obs", [](const Tetris &t) {
            ot_obs o;
            ot_pack_obs(&t.g, &o);
            return py::bytes(reinterpret_cast<const char *>(&o), sizeof(o));
        })
        .def("obs_hash", [](const Tetris &t) {
            ot_obs o;
            ot_pack_obs(&t.g, &o);
            return ot_hash_obs(&o);
        })
        .def("set_packed", [](Tetris &t, py::bytes b) {
            std::string s = b;
            if (s.size() != sizeof(t.g)) throw std::invalid_argument("need 64 bytes");
            std::memcpy(&t.g, s.data(), sizeof(t.g));
            t.sync();
        })
        .def_readonly("end", &Tetris::end)
        .def_readonly("score", &Tetris::score)
        .def_readonly("line_clears", &Tetris::line_clears)
        .def_readonly("combo", &Tetris::combo)
        .def_property_readonly("line_stats", [](const Tetris &t) {
            py::array_t<int32_t> a(4);
            for (int i = 0; i < 4; ++i) a.mutable_data()[i] = t.line_stats[i];
            return a;
        })
        .def("hash64", [](const Tetris &t) { return (uint64_t)t.hash(); })
        .def("__hash__", [](const Tetris &t) { return (py::ssize_t)(t.hash() & 0x7FFFFFFFFFFFFFFFULL); })
        .def("__eq__", [](const Tetris &a, const Tetris &b) { return a == b; }, py::is_operator());
    m.def("piece_at", [](uint32_t seed, uint32_t i, int randomizer) { return ot_piece_at(seed, i, randomizer); });
}
