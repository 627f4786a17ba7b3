/*
 * ORACLE — test infrastructure only.  Exports the three distribution helpers that the reference defines in
 * agents/cppmodule/core.h:387-449 (transform_distribution, mean_dist, mean_variance_dist) but never registers in
 * core.cpp:20-26.  The header is included where it lies under $(REF); nothing of it is copied here.
 * core.h defines non-inline functions, so it can be included by exactly one translation unit of a module: this one.
 */
#include "core.h"

PYBIND11_MODULE(dist, m) {
    m.def("transform_distribution", &transform_distribution);
    m.def("mean_dist", &mean_dist);
    m.def("mean_variance_dist", &mean_variance_dist);
}
