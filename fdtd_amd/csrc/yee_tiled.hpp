// Tuned large-grid kernels (x-marching tiles).  Placeholder: nothing applies yet.
#pragma once
#include "plan.hpp"
namespace fdtdhip {
template <typename T> inline bool tiled_applicable(const BulkArgs<T>&) { return false; }
template <typename T> inline int launch_tiled(const BulkArgs<T>&, bool, hipStream_t) { return -1; }
}  // namespace fdtdhip
