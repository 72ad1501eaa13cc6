// Half steps over a box of cells, out of place: the companions of the marching whole-step
// kernel (yee_march.hpp) on plans with PML faces.
//
// The marching kernel is fast because it carries no face code (with it, it was bound by
// instruction issue and slower than two half-step launches).  A plan with PMLs is therefore
// split: the box no face reaches (BulkArgs::core) is marched, and the shell around it -- at
// most six disjoint boxes, 11 % of the cells of a 512^3 grid with ten-cell faces -- takes two
// plain half steps with every feature of the half step (faces, objects), written to the same
// second pair of buffers:
//   1. shell_E over the shell boxes:   E1 = E half step of (E0, H0)   (memory variables in place)
//   2. step_march_box over the core:   E1, H1 of the core; the new E of the shell cells next to
//                                      the core is read from E1 where the whole-grid kernel
//                                      evaluates a row, plane or cell past its tile redundantly
//   3. shell_H over the shell boxes:   H1 = H half step of (H0, E1)
// Same operations on the same operands as two half steps everywhere: bitwise equal.
// (Requesting the memory variables of a shell cell together with its fields, before the curl,
// was measured: shell_E 0.27 -> 0.34 ms, shell_H 0.17 -> 0.19 ms at 512^3 with six ten-cell
// faces and two 64^3 boxes.  Not kept.  Neither was a variant that takes the faces as kernel
// arguments instead of walking the plan's table -- a lane finds the face of each axis by
// comparison and requests every memory variable up front, the form that halves the cost of
// the reverse sweep's shell launches, which evaluate four cells per lane: here, with one
// evaluation per lane, the shell of the 512^3 scene went from 0.396 to 0.492 ms per step and
// that of 256 x 1024 x 1024 from 0.716 to 0.921.)
#pragma once

#include "plan.hpp"
#include "yee_generic.hpp"
#include "yee_step.hpp"

namespace fdtdhip {

struct ShellBox { int i0, i1, j0, j1, k0, k1; };

// The boxes of the shell for one launch: box n takes the workgroups [first[n], first[n+1]).
constexpr int MAX_SHELL_BOXES = 7;   // six slabs around the core, the box of the objects
struct Shell {
  ShellBox box[MAX_SHELL_BOXES];
  int64_t first[MAX_SHELL_BOXES + 1];
  int count;
};

// Lane -> cell of its box, z fastest (a thin box along z gives a wave several short rows).
__device__ __forceinline__ bool shell_lane_cell(const Shell& s, int& i, int& j, int& k) {
  // (every member chosen as a value, box after box: a choice between the boxes themselves
  // becomes an address into the kernel argument and moves it to scratch memory)
  ShellBox b = s.box[0];
  int64_t first = s.first[0];
#pragma unroll
  for (int m = 1; m < MAX_SHELL_BOXES; ++m) {
    const bool pick = m < s.count && (int64_t)blockIdx.x >= s.first[m];
    b.i0 = pick ? s.box[m].i0 : b.i0; b.i1 = pick ? s.box[m].i1 : b.i1;
    b.j0 = pick ? s.box[m].j0 : b.j0; b.j1 = pick ? s.box[m].j1 : b.j1;
    b.k0 = pick ? s.box[m].k0 : b.k0; b.k1 = pick ? s.box[m].k1 : b.k1;
    first = pick ? s.first[m] : first;
  }
  const int64_t t = ((int64_t)blockIdx.x - first) * blockDim.x + threadIdx.x;
  const int nk = b.k1 - b.k0, nj = b.j1 - b.j0;
  const int64_t rows = t / nk;
  if (rows >= (int64_t)nj * (b.i1 - b.i0)) return false;
  k = b.k0 + (int)(t - rows * nk);
  i = b.i0 + (int)(rows / nj);
  j = b.j0 + (int)(rows - (int64_t)(i - b.i0) * nj);
  return true;
}

// Additive source points that fall on cells of the shell: added to its new E before the
// marching kernel and the shell's H half step read it (reference fdtd/grid.py:293-295).
template <typename T>
struct ShellSources {
  int n;
  int64_t elem[MAX_STEP_SOURCES];
  T value[MAX_STEP_SOURCES];
};
// One lane adds them one after the other, in the order of the plan's sources: two sources may
// sit on the same cell, and the reference's sums are sequential.
template <typename T>
__global__ void shell_sources(T* E1, ShellSources<T> s) {
  if (threadIdx.x != 0) return;
#pragma unroll
  for (int n = 0; n < MAX_STEP_SOURCES; ++n)
    if (n < s.n) E1[s.elem[n]] += s.value[n];
}

// E half step of the cells of `b`: a.F (read) -> out (which may be a.F itself: a lane reads its
// own cell of a.F only, before it writes).  The arithmetic of bulk_E_generic.
template <typename T, bool HAS_PML, bool HAS_OBJ>
__global__ __launch_bounds__(256) void shell_E(BulkArgs<T> a, T* out, Shell b) {
  int i, j, k;
  if (!shell_lane_cell(b, i, j, k)) return;
  const int64_t sz = 3, sy = 3 * (int64_t)a.nz, sx = sy * a.ny;
  const int64_t c3 = (int64_t)i * sx + (int64_t)j * sy + 3 * k;
  // every load requested before the first is used (a missing neighbour reads the cell itself
  // and its difference is dropped)
  const T* __restrict__ g = a.G + c3;
  const T* __restrict__ gx = i > 0 ? g - sx : g;
  const T* __restrict__ gy = j > 0 ? g - sy : g;
  const T* __restrict__ gz = k > 0 ? g - sz : g;
  // plane 0 of an x-slab takes the x neighbour's two components from the ghost plane
  const bool from_ghost = i == 0 && a.ghost != nullptr;
  const T* __restrict__ gh = from_ghost ? a.ghost + ((int64_t)j * a.nz + k) * 2 : gx + 1;
  const bool hasx = i > 0 || from_ghost;
  const T hx = g[0], hy = g[1], hz = g[2];
  const T xy = gh[0], xz = gh[1], yx = gy[0], yz = gy[2], zx = gz[0], zy = gz[1];
  T e[3] = {a.F[c3], a.F[c3 + 1], a.F[c3 + 2]};
  Diffs<T> d;
  d.y_z = j > 0 ? hz - yz : T(0); d.y_x = j > 0 ? hx - yx : T(0);
  d.z_y = k > 0 ? hy - zy : T(0); d.z_x = k > 0 ? hx - zx : T(0);
  d.x_z = hasx ? hz - xz : T(0); d.x_y = hasx ? hy - xy : T(0);
  T m[3];
  cell_material<T, false, HAS_OBJ>(a, c3, i, j, k, m);
  const T hs[3] = {hx, hy, hz};
  finish_E<T, HAS_PML, HAS_OBJ>(a, i, j, k, d, e, m, hs);
  out[c3] = e[0];
  out[c3 + 1] = e[1];
  out[c3 + 2] = e[2];
}

// H half step of the cells of `b`: a.F (read) -> out, driven by a.G (the new E).
template <typename T, bool HAS_PML>
__global__ __launch_bounds__(256) void shell_H(BulkArgs<T> a, T* out, Shell b) {
  int i, j, k;
  if (!shell_lane_cell(b, i, j, k)) return;
  const int64_t sz = 3, sy = 3 * (int64_t)a.nz, sx = sy * a.ny;
  const int64_t c3 = (int64_t)i * sx + (int64_t)j * sy + 3 * k;
  // the last plane of an x-slab takes the x neighbour's two components from the ghost plane
  const bool from_ghost = i == a.nx - 1 && a.ghost != nullptr;
  const bool ox = i < a.nx - 1 || from_ghost, oy = j < a.ny - 1, oz = k < a.nz - 1;
  const T* __restrict__ g = a.G + c3;
  const T* __restrict__ gx = i < a.nx - 1 ? g + sx : g;
  const T* __restrict__ gy = oy ? g + sy : g;
  const T* __restrict__ gz = oz ? g + sz : g;
  const T* __restrict__ gh = from_ghost ? a.ghost + ((int64_t)j * a.nz + k) * 2 : gx + 1;
  const T ex = g[0], ey = g[1], ez = g[2];
  const T xy = gh[0], xz = gh[1], yx = gy[0], yz = gy[2], zx = gz[0], zy = gz[1];
  T h[3] = {a.F[c3], a.F[c3 + 1], a.F[c3 + 2]};
  Diffs<T> d;
  d.y_z = oy ? yz - ez : T(0); d.y_x = oy ? yx - ex : T(0);
  d.z_y = oz ? zy - ey : T(0); d.z_x = oz ? zx - ex : T(0);
  d.x_z = ox ? xz - ez : T(0); d.x_y = ox ? xy - ey : T(0);
  T m[3];
  cell_material<T, false, false>(a, c3, i, j, k, m);
  finish_H<T, HAS_PML>(a, i, j, k, d, h, m);
  out[c3] = h[0];
  out[c3 + 1] = h[1];
  out[c3 + 2] = h[2];
}

}  // namespace fdtdhip
