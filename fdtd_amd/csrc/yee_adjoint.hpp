// Reverse-time (adjoint) Yee half steps for differentiable runs.
//
// The forward half steps are linear in the fields, so the adjoint recursion needs
// no forward field history; the only forward quantity it needs is the curl of H
// inside learnable anisotropic objects (for the gradient with respect to their
// inverse-permittivity tensors), which the forward E half step records per step
// (ObjK::trace).  This replaces the reference's autograd tape over every slice
// operation of every step (reference examples/learnable_focus.py:98-128 running
// through fdtd/objects.py:342-349).
//
// Each adjoint half step is two kernels:
//   pass 1 (per cell): adjoint of everything the forward bulk kernel did at that
//           cell after forming the six neighbour differences -> the adjoints of
//           those differences (work[cell][6]), the gradient contribution of
//           learnable objects, the adjoint of the PML memory variables;
//   pass 2 (per cell): transpose of the difference stencil as a gather from
//           `work`, added to the adjoint of the driving field.
// No atomics: results are reproducible bit for bit.
#pragma once

#include "plan.hpp"
#include "yee_tiled.hpp"

namespace fdtdhip {

// Cell of this lane: ROWS = the half-step row kernel's mapping (a wave per 64-cell z-row
// segment, 4 rows per workgroup, rotated segments: neighbours stay in the XCD's L2),
// otherwise one lane per cell of the flattened grid.
template <typename T, bool ROWS>
__device__ __forceinline__ bool adj_lane_cell(const AdjArgs<T>& a, int& i, int& j, int& k) {
  if (ROWS) return lane_cell<false>(0, a.ny, a.nz, a.rot_shift, i, j, k);
  const int64_t plane = (int64_t)a.ny * a.nz;
  const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= plane * a.nx) return false;
  i = (int)(cell / plane);
  const int64_t r = cell % plane;
  j = (int)(r / a.nz);
  k = (int)(r % a.nz);
  return true;
}

// Pass 1 over boxes of cells only (see adjoint_cells): a wave per 64-cell piece of a z row of
// a box, 4 rows per workgroup (thin boxes: see box_lane_cell).
struct CellBox { int i0, i1, j0, j1, k0, k1; };

// A box at most 16 cells deep along z (a z-face slab, a 2-D grid) gives each wave several
// rows of 1, 2, 4, 8 or 16 cells instead of a mostly empty 64-cell segment.
__host__ __device__ __forceinline__ int thin_box_shift(int depth) {
  return depth <= 1 ? 0 : (depth <= 2 ? 1 : (depth <= 4 ? 2 : (depth <= 8 ? 3 : 4)));
}

// The boxes of one pass-1 launch: box n takes the workgroups [first[n], first[n+1]) of the
// flat grid, gx[n] x gy[n] of them per x plane of the box (one launch instead of one per box:
// the boxes are small and each launch ended in a tail of its own; 256^3 float32 with six faces
// and a 64^3 box, reverse step 0.517 -> 0.482 ms; 3000 x 3000 x 1, 0.178 -> 0.156).
constexpr int MAX_CELL_BOXES = 7;   // six slabs around the core, the hull of the objects
struct CellBoxes {
  CellBox box[MAX_CELL_BOXES];
  int64_t first[MAX_CELL_BOXES + 1];
  int gx[MAX_CELL_BOXES], gy[MAX_CELL_BOXES];
  int count;
};

// Workgroups a box takes along z and y (per x plane).
__host__ __device__ __forceinline__ void box_grid(const CellBox& b, int& gx, int& gy) {
  const bool thin = b.k1 - b.k0 <= 16;   // several short rows per wave
  const int rows_per_block = thin ? 4 * (64 >> thin_box_shift(b.k1 - b.k0)) : 4;
  gx = thin ? 1 : (b.k1 - b.k0 + 63) / 64;
  gy = (b.j1 - b.j0 + rows_per_block - 1) / rows_per_block;
}

__device__ __forceinline__ bool box_lane_cell(const CellBoxes& t, int& i, int& j, int& k) {
  // (every member chosen as a value, box after box: a choice between the boxes themselves
  // becomes an address into the kernel argument and moves it to scratch memory)
  CellBox b = t.box[0];
  int64_t first = t.first[0];
  int gx = t.gx[0], gy = t.gy[0];
#pragma unroll
  for (int m = 1; m < MAX_CELL_BOXES; ++m) {
    const bool pick = m < t.count && (int64_t)blockIdx.x >= t.first[m];
    b.i0 = pick ? t.box[m].i0 : b.i0; b.i1 = pick ? t.box[m].i1 : b.i1;
    b.j0 = pick ? t.box[m].j0 : b.j0; b.j1 = pick ? t.box[m].j1 : b.j1;
    b.k0 = pick ? t.box[m].k0 : b.k0; b.k1 = pick ? t.box[m].k1 : b.k1;
    first = pick ? t.first[m] : first;
    gx = pick ? t.gx[m] : gx;
    gy = pick ? t.gy[m] : gy;
  }
  const int64_t local = (int64_t)blockIdx.x - first;
  const int bx = (int)(local % gx);
  const int64_t rest = local / gx;
  const int by = (int)(rest % gy), bz = (int)(rest / gy);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  i = b.i0 + bz;
  if (b.k1 - b.k0 <= 16) {
    const int sh = thin_box_shift(b.k1 - b.k0);
    k = b.k0 + (lane & ((1 << sh) - 1));
    j = b.j0 + (by * 4 + wave) * (64 >> sh) + (lane >> sh);
  } else {
    k = b.k0 + bx * 64 + lane;
    j = b.j0 + by * 4 + wave;
  }
  return i < b.i1 && k < b.k1 && j < b.j1;
}

// order of the six difference adjoints in `work`
enum { WX_Y = 0, WX_Z = 1, WY_X = 2, WY_Z = 3, WZ_X = 4, WZ_Y = 5 };

template <typename T, int AXIS>
__device__ __forceinline__ void adj_pml_face(const PmlK<T>& P, int nx, int i, int j, int k,
                                             bool phaseE, T sc, const T lam[3], const T m[3],
                                             T& GdFu, T& GdFv, bool store = true) {
  constexpr int u = (AXIS + 1) % 3, v = (AXIS + 2) % 3;
  const int coord = AXIS == 0 ? i : (AXIS == 1 ? j : k);
  const int l = coord - P.lo;
  if (l < 0 || l >= P.t) return;
  const int li = AXIS == 0 ? l : i, lj = AXIS == 1 ? l : j, lk = AXIS == 2 ? l : k;
  const int64_t s = ((int64_t)li * P.sy + lj) * P.sz + lk;
  const int64_t slab = (int64_t)(AXIS == 0 ? P.t : nx) * P.sy * P.sz;
  T lp0 = P.psi[s], lp1 = P.psi[slab + s];  // adjoint of the updated memory variables
  if (P.fused) {
    // forward: F_u += sign*(sc*m_u)*(-psi0'), F_v += sign*(sc*m_v)*psi1', sign = +1 (E) / -1 (H)
    const T sg = phaseE ? T(1) : T(-1);
    lp0 += sg * (-(sc * m[u])) * lam[u];
    lp1 += sg * (sc * m[v]) * lam[v];
  }
  // forward: psi0' = b*psi0 + gate*c*dFv ; psi1' = b*psi1 + gate*c*dFu
  if (phaseE ? (l >= 1) : (l <= P.t - 2)) {
    GdFv += P.c[l] * lp0;
    GdFu += P.c[l] * lp1;
  }
  if (store) {
    P.psi_out[s] = P.b[l] * lp0;
    P.psi_out[slab + s] = P.b[l] * lp1;
  }
}

// Inverse material of a cell for the adjoint kernels: the proven constant (zero under an
// object) or the array entry, exactly as the forward kernels took it.
template <typename T>
__device__ __forceinline__ void adj_material(const AdjArgs<T>& a, int64_t c3, int i, int j, int k,
                                             T m[3]) {
  if (a.mat_const) {
    bool covered = false;
    const bool near = i >= a.hull[0] && i < a.hull[1] && j >= a.hull[2] && j < a.hull[3] &&
                      k >= a.hull[4] && k < a.hull[5];
    if (a.nobj && __any(near))
      for (int o = 0; o < a.nobj; ++o) {
        const ObjK<T> O = load_obj<T>(a, o);
        covered |= i >= O.x0 && i < O.x1 && j >= O.y0 && j < O.y1 && k >= O.z0 && k < O.z1;
      }
#pragma unroll
    for (int c = 0; c < 3; ++c) m[c] = covered ? T(0) : a.bg[c];
  } else {
    m[0] = a.mat[c3]; m[1] = a.mat[c3 + 1]; m[2] = a.mat[c3 + 2];
  }
}

// Wave-uniform tests: does any lane's cell lie outside the PML-free core / inside the
// hull of the objects?  (Same skips as the forward kernels.)
template <typename T>
__device__ __forceinline__ bool adj_any_pml(const AdjArgs<T>& a, int i, int j, int k) {
  const bool in_core = i >= a.core[0] && i < a.core[1] && j >= a.core[2] && j < a.core[3] &&
                       k >= a.core[4] && k < a.core[5];
  return a.npml > 0 && !__all(in_core);
}
// Whether some lane's cell lies in PML face p (the ranges are kernel arguments, already in
// scalar registers: a wave does not wait for the descriptors of faces it does not touch).
template <typename T>
__device__ __forceinline__ bool adj_face_touched(const AdjArgs<T>& a, int p, int i, int j, int k) {
  const int ax = a.face_axis[p];
  const int coord = ax == 0 ? i : (ax == 1 ? j : k);
  return __any(coord >= a.face_lo[p] && coord < a.face_hi[p]);
}
template <typename T>
__device__ __forceinline__ bool adj_any_obj(const AdjArgs<T>& a, int i, int j, int k) {
  const bool near = i >= a.hull[0] && i < a.hull[1] && j >= a.hull[2] && j < a.hull[3] &&
                    k >= a.hull[4] && k < a.hull[5];
  return a.nobj > 0 && __any(near);
}

// "Plain" cells: outside every PML slab and outside the hull of the objects, with the
// material a proven constant.  There pass 1 is a fixed linear map of the cell's own lamF
// (no memory variable, no object, lamF left as it is), so pass 2 can evaluate it for the
// cell and the three neighbours it gathers from instead of reading `work`: 9 values per
// cell and half step through memory instead of 21.  The choice is made per cell: pass 2
// takes the direct form where the cell and the neighbours it reaches are plain, and pass 1
// leaves a cell out when every cell that could read its entry -- itself and its six
// neighbours' gathers, all of which reach at most one cell further -- takes the direct
// form, i.e. when the 3 x 3 x 3 block around it is plain.  Inclusive cell ranges, clamped
// to the grid -- except across the open faces of an x-slab, where nothing is plain: the
// planes next to such a face keep the two-pass form, and with it the planes of `work` the
// ranks exchange.
template <typename T>
__device__ __forceinline__ bool adj_box_plain(const AdjArgs<T>& a, int i0, int i1, int j0, int j1,
                                              int k0, int k1) {
  // across an open face of an x-slab lies another rank's plane: its difference adjoints
  // arrive as a plane of `work`, and this rank's plane at the face is sent as one
  if ((a.open_lo && i0 < 0) || (a.open_hi && i1 > a.nx - 1)) return false;
  i0 = max(i0, 0); j0 = max(j0, 0); k0 = max(k0, 0);
  i1 = min(i1, a.nx - 1); j1 = min(j1, a.ny - 1); k1 = min(k1, a.nz - 1);
  const bool in_core = i0 >= a.core[0] && i1 < a.core[1] && j0 >= a.core[2] && j1 < a.core[3] &&
                       k0 >= a.core[4] && k1 < a.core[5];
  const bool meets_hull = a.nobj > 0 && i1 >= a.hull[0] && i0 < a.hull[1] && j1 >= a.hull[2] &&
                          j0 < a.hull[3] && k1 >= a.hull[4] && k0 < a.hull[5];
  return in_core && !meets_hull;
}

// Pass 1 has nothing to leave behind for this cell.
template <typename T>
__device__ __forceinline__ bool adj_cells_skippable(const AdjArgs<T>& a, int i, int j, int k) {
  return a.direct && adj_box_plain(a, i - 1, i + 1, j - 1, j + 1, k - 1, k + 1);
}

// The difference adjoints of a plain cell from its lamF, as pass 1 would have stored them
// (edge rule included: differences the forward pass dropped carry no adjoint).
template <typename T, bool PHASE_E>
__device__ __forceinline__ void adj_plain_G(const AdjArgs<T>& a, const T* lam, int i, int j, int k,
                                            T G[6]) {
  T g[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const T v = (a.sc * a.bg[c]) * lam[c];
    g[c] = PHASE_E ? v : -v;
  }
  G[WY_Z] = g[0]; G[WZ_Y] = -g[0];
  G[WZ_X] = g[1]; G[WX_Z] = -g[1];
  G[WX_Y] = g[2]; G[WY_X] = -g[2];
  if (PHASE_E) {
    if (i == 0 && !a.open_lo) G[WX_Y] = G[WX_Z] = 0;
    if (j == 0) G[WY_X] = G[WY_Z] = 0;
    if (k == 0) G[WZ_X] = G[WZ_Y] = 0;
  } else {
    if (i == a.nx - 1 && !a.open_hi) G[WX_Y] = G[WX_Z] = 0;
    if (j == a.ny - 1) G[WY_X] = G[WY_Z] = 0;
    if (k == a.nz - 1) G[WZ_X] = G[WZ_Y] = 0;
  }
}

// Pass 2 of a plain cell with plain neighbours: the same sums as
// adj_gather_cell, the difference adjoints taken from lamF (a.lamF is not written by
// anything while the two passes of a half step run, outside absorbing objects).
template <typename T, bool PHASE_E>
__device__ __forceinline__ void adj_gather_direct(const AdjArgs<T>& a, int i, int j, int k, T add[3]) {
  const int64_t sy = 3 * (int64_t)a.nz, sx = sy * a.ny;
  const T* l = a.lamF + (((int64_t)i * a.ny + j) * a.nz + k) * 3;
  const int s = PHASE_E ? 1 : -1;
  const bool ox = PHASE_E ? i < a.nx - 1 : i > 0;
  const bool oy = PHASE_E ? j < a.ny - 1 : j > 0;
  const bool oz = PHASE_E ? k < a.nz - 1 : k > 0;
  // all loads first (a missing neighbour reads the cell itself and is discarded)
  const T* lx = ox ? l + s * sx : l;
  const T* ly = oy ? l + s * sy : l;
  const T* lz = oz ? l + s * 3 : l;
  const T own[3] = {l[0], l[1], l[2]};
  const T vx[3] = {lx[0], lx[1], lx[2]};
  const T vy[3] = {ly[0], ly[1], ly[2]};
  const T vz[3] = {lz[0], lz[1], lz[2]};
  T w[6], wx[6], wy[6], wz[6];
  adj_plain_G<T, PHASE_E>(a, own, i, j, k, w);
  adj_plain_G<T, PHASE_E>(a, vx, i + s, j, k, wx);
  adj_plain_G<T, PHASE_E>(a, vy, i, j + s, k, wy);
  adj_plain_G<T, PHASE_E>(a, vz, i, j, k + s, wz);
  const T zero = T(0);
  if (PHASE_E) {
    add[0] = (w[WY_X] - (oy ? wy[WY_X] : zero)) + (w[WZ_X] - (oz ? wz[WZ_X] : zero));
    add[1] = (w[WX_Y] - (ox ? wx[WX_Y] : zero)) + (w[WZ_Y] - (oz ? wz[WZ_Y] : zero));
    add[2] = (w[WX_Z] - (ox ? wx[WX_Z] : zero)) + (w[WY_Z] - (oy ? wy[WY_Z] : zero));
  } else {
    add[0] = ((oy ? wy[WY_X] : zero) - w[WY_X]) + ((oz ? wz[WZ_X] : zero) - w[WZ_X]);
    add[1] = ((ox ? wx[WX_Y] : zero) - w[WX_Y]) + ((oz ? wz[WZ_Y] : zero) - w[WZ_Y]);
    add[2] = ((ox ? wx[WX_Z] : zero) - w[WX_Z]) + ((oy ? wy[WY_Z] : zero) - w[WY_Z]);
  }
}

// The adjoint of everything the forward E half step did at one cell after forming the six
// neighbour differences: on return G holds the adjoints of those differences and lam the
// adjoint of the cell's E before the half step.  store = false evaluates the cell without
// side effects (memory-variable adjoints, object gradients), for a lane that needs a
// neighbour's G.  Returns whether lam changed (only absorbing objects rescale it).
// ONLY_AXIS >= 0: only the entries of G for differences along that axis are wanted (a lane
// evaluating a neighbour it gathers from): PML faces normal to the other axes add to other
// entries and are left out.
template <typename T, int ONLY_AXIS = -1>
__device__ __forceinline__ bool adj_E_cell(const AdjArgs<T>& a, int i, int j, int k, int64_t c3,
                                           T lam[3], T G[6], bool store) {
  T m[3];
  adj_material<T>(a, c3, i, j, k, m);
#pragma unroll
  for (int n = 0; n < 6; ++n) G[n] = T(0);
  bool lam_changed = false;

  // PML corrections and memory variables (they act on the final E of the half step)
  if (adj_any_pml(a, i, j, k))
#pragma unroll
  for (int q = 0; q < MAX_PML; ++q) {   // last registered first
    const int p = MAX_PML - 1 - q;
    if (p >= a.npml) continue;
    if (ONLY_AXIS >= 0 && a.face_axis[p] != ONLY_AXIS) continue;
    if (!adj_face_touched(a, p, i, j, k)) continue;
    const PmlK<T> P = a.pml[p];
    if (P.axis == 0) adj_pml_face<T, 0>(P, a.nx, i, j, k, true, a.sc, lam, m, G[WX_Y], G[WX_Z], store);
    else if (P.axis == 1) adj_pml_face<T, 1>(P, a.nx, i, j, k, true, a.sc, lam, m, G[WY_Z], G[WY_X], store);
    else adj_pml_face<T, 2>(P, a.nx, i, j, k, true, a.sc, lam, m, G[WZ_X], G[WZ_Y], store);
  }

  // objects, last registered first
  T g[3] = {0, 0, 0};
  if (adj_any_obj(a, i, j, k))
  for (int o = a.nobj - 1; o >= 0; --o) {
    const ObjK<T> O = load_obj<T>(a, o);
    if (i < O.x0 || i >= O.x1 || j < O.y0 || j >= O.y1 || k < O.z0 || k >= O.z1) continue;
    const int64_t oc =
        ((int64_t)(i - O.x0) * (O.y1 - O.y0) + (j - O.y0)) * (O.z1 - O.z0) + (k - O.z0);
    if (O.kind == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] += (a.sc * O.inv[oc * 3 + c]) * lam[c];
    } else if (O.kind == 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const T f = O.absorb[oc * 3 + c];
        g[c] += ((a.sc * O.inv[oc * 3 + c]) / (T(1) + f)) * lam[c];
        lam[c] *= (T(1) - f) / (T(1) + f);
      }
      lam_changed = true;
    } else if (O.kind == 3) {
      // forward: E_r += sc * sum_c (M_rc * mod_rc) * cu_c with mod from the traced energies,
      // which are constants of the step (the reference takes them without gradient,
      // fdtd/objects.py:329-333)
      if (!O.etrace || !O.trace || a.trace_step < 0) continue;  // refused on the host
      const int64_t box = (int64_t)(O.x1 - O.x0) * (O.y1 - O.y0) * (O.z1 - O.z0);
      const T* et = O.etrace + (a.trace_step * box + oc) * 2;
      const T* cu = O.trace + (a.trace_step * box + oc) * 3;
      const T eE = et[0], eH = et[1];
      const T* M = O.inv + oc * 9;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int dcol = 0; dcol < 3; ++dcol) {
          const T* w = O.modw + (rr * 3 + dcol) * 3;
          const T mod = T(1) / (T(1) + exp(-(w[0] * eE + w[1] * eH + w[2])));
          g[dcol] += a.sc * ((M[rr * 3 + dcol] * mod) * lam[rr]);
          const T outer = (a.sc * lam[rr]) * cu[dcol];  // dL/d(M_rc * mod_rc)
          if (store && O.grad) O.grad[oc * 9 + rr * 3 + dcol] += outer * mod;
          if (store && O.gmod) {
            const T dz = (outer * M[rr * 3 + dcol]) * (mod * (T(1) - mod));
            T* gm = O.gmod + (oc * 9 + rr * 3 + dcol) * 3;
            gm[0] += dz * eE;
            gm[1] += dz * eH;
            gm[2] += dz;
          }
        }
    } else {
      const T* M = O.inv + oc * 9;
#pragma unroll
      for (int dcol = 0; dcol < 3; ++dcol) {
        T s = M[0 * 3 + dcol] * lam[0];
        s += M[1 * 3 + dcol] * lam[1];
        s += M[2 * 3 + dcol] * lam[2];
        g[dcol] += a.sc * s;
      }
      if (store && O.grad && O.trace && a.trace_step >= 0) {
        const int64_t box = (int64_t)(O.x1 - O.x0) * (O.y1 - O.y0) * (O.z1 - O.z0);
        const T* cu = O.trace + (a.trace_step * box + oc) * 3;
        T* gr = O.grad + oc * 9;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int dcol = 0; dcol < 3; ++dcol) gr[rr * 3 + dcol] += (a.sc * lam[rr]) * cu[dcol];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) g[c] += (a.sc * m[c]) * lam[c];

  // curl: cu0 = d.y_z - d.z_y, cu1 = d.z_x - d.x_z, cu2 = d.x_y - d.y_x
  G[WY_Z] += g[0]; G[WZ_Y] -= g[0];
  G[WZ_X] += g[1]; G[WX_Z] -= g[1];
  G[WX_Y] += g[2]; G[WY_X] -= g[2];
  // differences the forward pass dropped at the low edges carry no adjoint
  if (i == 0 && !a.open_lo) G[WX_Y] = G[WX_Z] = 0;
  if (j == 0) G[WY_X] = G[WY_Z] = 0;
  if (k == 0) G[WZ_X] = G[WZ_Y] = 0;
  return lam_changed;
}

// Pass 1 of the adjoint of the E half step.
template <typename T, bool ROWS, bool BOXED = false>
__global__ __launch_bounds__(256) void adj_E_cells(AdjArgs<T> a, CellBoxes box) {
  int i, j, k;
  if (BOXED ? !box_lane_cell(box, i, j, k) : !adj_lane_cell<T, ROWS>(a, i, j, k)) return;
  if (adj_cells_skippable(a, i, j, k)) return;
  const int64_t cell = ((int64_t)i * a.ny + j) * a.nz + k;
  const int64_t c3 = cell * 3;
  T lam[3] = {a.lamF[c3], a.lamF[c3 + 1], a.lamF[c3 + 2]};
  T G[6];
  const bool lam_changed = adj_E_cell<T>(a, i, j, k, c3, lam, G, true);
  T* w = a.work + cell * 6;
#pragma unroll
  for (int n = 0; n < 6; ++n) w[n] = G[n];
  if (lam_changed) {  // only absorbing objects rescale the field adjoint
    a.lamF[c3] = lam[0];
    a.lamF[c3 + 1] = lam[1];
    a.lamF[c3 + 2] = lam[2];
  }
}

// The same for the H half step (no objects act in that half step).
template <typename T, int ONLY_AXIS = -1>
__device__ __forceinline__ void adj_H_cell(const AdjArgs<T>& a, int i, int j, int k, int64_t c3,
                                           const T lam[3], T G[6], bool store) {
  T m[3];
  adj_material<T>(a, c3, i, j, k, m);
#pragma unroll
  for (int n = 0; n < 6; ++n) G[n] = T(0);
  if (adj_any_pml(a, i, j, k))
#pragma unroll
  for (int q = 0; q < MAX_PML; ++q) {   // last registered first
    const int p = MAX_PML - 1 - q;
    if (p >= a.npml) continue;
    if (ONLY_AXIS >= 0 && a.face_axis[p] != ONLY_AXIS) continue;
    if (!adj_face_touched(a, p, i, j, k)) continue;
    const PmlK<T> P = a.pml[p];
    if (P.axis == 0) adj_pml_face<T, 0>(P, a.nx, i, j, k, false, a.sc, lam, m, G[WX_Y], G[WX_Z], store);
    else if (P.axis == 1) adj_pml_face<T, 1>(P, a.nx, i, j, k, false, a.sc, lam, m, G[WY_Z], G[WY_X], store);
    else adj_pml_face<T, 2>(P, a.nx, i, j, k, false, a.sc, lam, m, G[WZ_X], G[WZ_Y], store);
  }
  T g[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) g[c] = -((a.sc * m[c]) * lam[c]);  // H' = H - (sc*mu)*curl
  G[WY_Z] += g[0]; G[WZ_Y] -= g[0];
  G[WZ_X] += g[1]; G[WX_Z] -= g[1];
  G[WX_Y] += g[2]; G[WY_X] -= g[2];
  if (i == a.nx - 1 && !a.open_hi) G[WX_Y] = G[WX_Z] = 0;
  if (j == a.ny - 1) G[WY_X] = G[WY_Z] = 0;
  if (k == a.nz - 1) G[WZ_X] = G[WZ_Y] = 0;
}

// Pass 1 of the adjoint of the H half step.
template <typename T, bool ROWS, bool BOXED = false>
__global__ __launch_bounds__(256) void adj_H_cells(AdjArgs<T> a, CellBoxes box) {
  int i, j, k;
  if (BOXED ? !box_lane_cell(box, i, j, k) : !adj_lane_cell<T, ROWS>(a, i, j, k)) return;
  if (adj_cells_skippable(a, i, j, k)) return;
  const int64_t cell = ((int64_t)i * a.ny + j) * a.nz + k;
  const int64_t c3 = cell * 3;
  const T lam[3] = {a.lamF[c3], a.lamF[c3 + 1], a.lamF[c3 + 2]};
  T G[6];
  adj_H_cell<T>(a, i, j, k, c3, lam, G, true);
  T* w = a.work + cell * 6;
#pragma unroll
  for (int n = 0; n < 6; ++n) w[n] = G[n];
}

// Pass 2: transpose of the difference stencil.  (A single-pass form, in which every lane
// re-derives the difference adjoints of the three neighbours its gather reaches and `work`
// disappears, was measured: 0.84 ms against 0.79 ms per reverse step at 256^3 -- the four
// evaluations per cell cost more issue slots than the saved traffic returns -- and dropped.)
//   E half step: d_a_w(x) = H_w(x) - H_w(x - e_a)  =>  lamH_w(x) += G_a_w(x) - G_a_w(x + e_a)
//   H half step: d_a_w(x) = E_w(x + e_a) - E_w(x)  =>  lamE_w(x) += G_a_w(x - e_a) - G_a_w(x)
template <typename T, bool PHASE_E>
__device__ __forceinline__ void adj_gather_cell(const AdjArgs<T>& a, int i, int j, int k,
                                                T add[3]) {
  const int64_t plane = (int64_t)a.ny * a.nz;
  const int64_t r = (int64_t)j * a.nz + k;
  const int64_t cell = (int64_t)i * plane + r;
  const T* w = a.work + cell * 6;
  const int64_t sx = plane * 6, sy = (int64_t)a.nz * 6, sz = 6;
  auto nb = [&](bool ok, int64_t off, int which) { return ok ? w[off + which] : T(0); };
  // x neighbour: inside the slab, or the neighbour rank's plane, or nothing at a domain edge
  auto nbx = [&](bool inside, int64_t off, bool open, int which) {
    if (inside) return w[off + which];
    return open && a.wghost ? a.wghost[r * 2 + which] : T(0);   // (WX_Y, WX_Z are entries 0 and 1)
  };
  if (PHASE_E) {
    const bool ox = i < a.nx - 1, oy = j < a.ny - 1, oz = k < a.nz - 1;
    add[0] = (w[WY_X] - nb(oy, sy, WY_X)) + (w[WZ_X] - nb(oz, sz, WZ_X));
    add[1] = (w[WX_Y] - nbx(ox, sx, a.open_hi, WX_Y)) + (w[WZ_Y] - nb(oz, sz, WZ_Y));
    add[2] = (w[WX_Z] - nbx(ox, sx, a.open_hi, WX_Z)) + (w[WY_Z] - nb(oy, sy, WY_Z));
  } else {
    const bool ox = i > 0, oy = j > 0, oz = k > 0;
    add[0] = (nb(oy, -sy, WY_X) - w[WY_X]) + (nb(oz, -sz, WZ_X) - w[WZ_X]);
    add[1] = (nbx(ox, -sx, a.open_lo, WX_Y) - w[WX_Y]) + (nb(oz, -sz, WZ_Y) - w[WZ_Y]);
    add[2] = (nbx(ox, -sx, a.open_lo, WX_Z) - w[WX_Z]) + (nb(oy, -sy, WY_Z) - w[WY_Z]);
  }
}

template <typename T, bool PHASE_E, bool ROWS>
__global__ __launch_bounds__(256) void adj_gather(AdjArgs<T> a) {
  int i, j, k;
  if (!adj_lane_cell<T, ROWS>(a, i, j, k)) return;
  const int64_t cell = ((int64_t)i * a.ny + j) * a.nz + k;
  T* out = a.lamG + cell * 3;
  // The adjoint being completed is requested first, so that it travels together with the
  // loads of the gather instead of after them (the choice between the two forms below is a
  // branch the compiler does not move a load across).  Streamed once per launch: in float32
  // non-temporal, to keep it out of the way of the neighbour re-reads (as the forward row
  // kernels do for the advanced field; float64 cells gain nothing from it).
  constexpr bool NT = ROWS && sizeof(T) == 4;
  T o[3];
  load3<T, NT>(out, o[0], o[1], o[2]);
  T add[3];
  if (a.direct && (PHASE_E ? adj_box_plain(a, i, i + 1, j, j + 1, k, k + 1)
                           : adj_box_plain(a, i - 1, i, j - 1, j, k - 1, k)))
    adj_gather_direct<T, PHASE_E>(a, i, j, k, add);
  else
    adj_gather_cell<T, PHASE_E>(a, i, j, k, add);
  store3<T, NT>(out, o[0] + add[0], o[1] + add[1], o[2] + add[2]);
}

// Small grids (one wave per SIMD, time = number of launches): pass 2 of one half step and
// pass 1 of the half step undone next, in one launch.  The gather completes the cell's
// adjoint of the field that the next pass 1 starts from, and pass 1 is local to the cell,
// so the only new requirement is that the two passes use different `work` arrays (other
// lanes still gather from the one being read).  g: arguments of the gather (its lamG is
// the field adjoint completed here); c: arguments of the pass 1 that follows (c.lamF is
// that same array, c.work the other work array).  GATHER_E: the gather belongs to the
// adjoint of the E half step and the pass 1 to that of the H half step of the step before.
template <typename T, bool GATHER_E>
__global__ __launch_bounds__(256) void adj_gather_then_cells(AdjArgs<T> g, AdjArgs<T> c) {
  int i, j, k;
  if (!adj_lane_cell<T, false>(g, i, j, k)) return;
  const int64_t cell = ((int64_t)i * g.ny + j) * g.nz + k;
  const int64_t c3 = cell * 3;
  T add[3];
  adj_gather_cell<T, GATHER_E>(g, i, j, k, add);
  T lam[3] = {g.lamG[c3], g.lamG[c3 + 1], g.lamG[c3 + 2]};
#pragma unroll
  for (int n = 0; n < 3; ++n) lam[n] += add[n];
  T G[6];
  if (GATHER_E) adj_H_cell<T>(c, i, j, k, c3, lam, G, true);
  else adj_E_cell<T>(c, i, j, k, c3, lam, G, true);
  T* w = c.work + cell * 6;
#pragma unroll
  for (int n = 0; n < 6; ++n) w[n] = G[n];
  c.lamF[c3] = lam[0];
  c.lamF[c3 + 1] = lam[1];
  c.lamF[c3 + 2] = lam[2];
}

// Adjoint of pml_apply_slab (a PML whose correction follows a periodic copy):
// only the memory-variable adjoints change; the recursion itself is undone in pass 1.
template <typename T>
__global__ void adj_pml_apply_slab(const T* lamF, const T* mat, PmlK<T> P, int nx, int ny, int nz,
                                   T sc, T sign) {
  const int sx = P.axis == 0 ? P.t : nx;
  const int64_t slab = (int64_t)sx * P.sy * P.sz;
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= slab) return;
  const int lk = (int)(s % P.sz);
  const int lj = (int)((s / P.sz) % P.sy);
  const int li = (int)(s / ((int64_t)P.sz * P.sy));
  const int i = P.axis == 0 ? li + P.lo : li, j = P.axis == 1 ? lj + P.lo : lj,
            k = P.axis == 2 ? lk + P.lo : lk;
  const int64_t c3 = (((int64_t)i * ny + j) * nz + k) * 3;
  const int u = (P.axis + 1) % 3, v = (P.axis + 2) % 3;
  P.psi[s] += sign * (-(sc * mat[c3 + u])) * lamF[c3 + u];
  P.psi[slab + s] += sign * (sc * mat[c3 + v]) * lamF[c3 + v];
}

// Adjoint of periodic_copy F[dst] = F[src]:  lam[src] += lam[dst]; lam[dst] = 0.
template <typename T>
__global__ void adj_periodic_copy(T* lam, int nx, int ny, int nz, int axis, int dst, int src) {
  const int n1 = axis == 0 ? ny : nx;
  const int n2 = axis == 2 ? ny : nz;
  const int64_t total = (int64_t)n1 * n2 * 3;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c = (int)(t % 3);
  const int b = (int)((t / 3) % n2);
  const int aa = (int)(t / (3 * (int64_t)n2));
  int64_t d, s;
  if (axis == 0) {
    d = (((int64_t)dst * ny + aa) * nz + b) * 3 + c;
    s = (((int64_t)src * ny + aa) * nz + b) * 3 + c;
  } else if (axis == 1) {
    d = (((int64_t)aa * ny + dst) * nz + b) * 3 + c;
    s = (((int64_t)aa * ny + src) * nz + b) * 3 + c;
  } else {
    d = (((int64_t)aa * ny + b) * nz + dst) * 3 + c;
    s = (((int64_t)aa * ny + b) * nz + src) * 3 + c;
  }
  lam[s] += lam[d];
  lam[d] = 0;
}

// Adjoint of source_modulated: the source adds to E without reading a differentiable
// quantity of the fields (the energies carry no gradient, as in the reference), so the
// field adjoint passes through; what is collected are the gradients of the source's own
// inputs.  One workgroup per channel c, lanes over the (x, y) columns of the box:
//   gP[c][column]  += lam * osc[c] * mod          (every step)
//   gosc[c]         = sum lam * P * mod           (this step's row)
//   gwb[c][0..2]   += sum dz * {eE, eH, 1},  dz = lam * osc[c] * P * mod * (1 - mod)
// Sums over cells go through a fixed LDS tree: reproducible.
// Adjoint of the voltage source fed back from its loop current (source_soft_point with
// detect_current of yee_generic.hpp; reference fdtd/sources.py:719-734 and
// fdtd/detectors.py:443-464): E[cell, z] took impedance * I / spacing, and I is the linear
// form of H around the cell that detect_current evaluates -- with the first layer's Hy
// difference counted in both layers and Python's wrap-around for index -1.  One lane;
// coinciding cells (an extent of one) just receive several of the terms.
// `below`: on an x-slab whose plane 0 holds the cell, the plane x - 1 belongs to another rank
// (the one below, or -- wrapping around -- the last one): its terms are ADDED to this
// (ny, nz, 3) plane instead, which the caller sends there.
template <typename T>
__global__ void adj_source_soft_point(const T* lamE, T* lamH, int64_t elem, int nx, int ny, int nz,
                                      double impedance, double spacing, T* below) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const int64_t cell = elem / 3;
  const int pz = (int)(cell % nz), py = (int)((cell / nz) % ny), px = (int)(cell / ((int64_t)nz * ny));
  const int xm = px == 0 ? nx - 1 : px - 1;
  const int ym = py == 0 ? ny - 1 : py - 1;
  const int zm = pz == 0 ? nz - 1 : pz - 1;
  const bool away = px == 0 && below != nullptr;
  auto at = [&](int i, int j, int k, int c) -> T& {
    return lamH[(((int64_t)i * ny + j) * nz + k) * 3 + c];
  };
  // the cell at plane x - 1 (told apart by name, not by index: on a slab of one plane xm == px)
  auto at_xm = [&](int j, int k, int c) -> T& {
    if (away) return below[((int64_t)j * nz + k) * 3 + c];
    return lamH[(((int64_t)xm * ny + j) * nz + k) * 3 + c];
  };
  // adjoint of the current, then of each of its two layers (I = (cur1 + cur2) / 2), times
  // the spacing every difference was scaled with
  const T g = (T)((double)lamE[elem] * (impedance / spacing));
  const T h = g / T(2) * (T)spacing;
  at(px, ym, pz, 0) += h; at(px, py, pz, 0) -= h;            // layer pz, Hx
  at(px, py, pz, 1) += h + h; at_xm(py, pz, 1) -= h + h;     // layer pz, Hy: in both currents
  at(px, ym, zm, 0) += h; at(px, py, zm, 0) -= h;            // layer pz - 1, Hx
  at(px, py, zm, 1) += h; at_xm(py, zm, 1) -= h;             // layer pz - 1, Hy
}

template <typename T>
__global__ __launch_bounds__(256) void adj_source_modulated(
    const T* lamE, int ny, int nz, int x0, int x1, int y0, int y1, int z0, int z1, int comp,
    const T* __restrict__ P, const T* __restrict__ osc, const T* __restrict__ wb,
    const T* __restrict__ etrace, T* gP, T* gosc, T* gwb) {
  const int c = blockIdx.x;
  const int bz = z1 - z0, by = y1 - y0;
  const int64_t cols = (int64_t)(x1 - x0) * by;
  const T w0 = wb[c * 3], w1 = wb[c * 3 + 1], w2 = wb[c * 3 + 2], oc = osc[c];
  T acc[4] = {0, 0, 0, 0};  // d osc, d w_E, d w_H, d bias
  for (int64_t col = threadIdx.x; col < cols; col += blockDim.x) {
    const int j = y0 + (int)(col % by), i = x0 + (int)(col / by);
    const T pc = P[c * cols + col];
    T gp = T(0);
    for (int lk = 0; lk < bz; ++lk) {
      const int64_t t = col * bz + lk;
      const T lam = lamE[(((int64_t)i * ny + j) * nz + z0 + lk) * 3 + comp];
      const T eE = etrace[t * 2], eH = etrace[t * 2 + 1];
      const T mod = T(1) / (T(1) + exp(-(w0 * eE + w1 * eH + w2)));
      gp += lam * mod;
      acc[0] += (lam * pc) * mod;
      const T dz = ((lam * oc) * pc) * (mod * (T(1) - mod));
      acc[1] += dz * eE;
      acc[2] += dz * eH;
      acc[3] += dz;
    }
    gP[c * cols + col] += gp * oc;
  }
  __shared__ T red[4][256];
#pragma unroll
  for (int n = 0; n < 4; ++n) red[n][threadIdx.x] = acc[n];
  __syncthreads();
  for (int half = 128; half > 0; half >>= 1) {
    if ((int)threadIdx.x < half)
#pragma unroll
      for (int n = 0; n < 4; ++n) red[n][threadIdx.x] += red[n][threadIdx.x + half];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    gosc[c] = red[0][0];
    gwb[c * 3] += red[1][0];
    gwb[c * 3 + 1] += red[2][0];
    gwb[c * 3 + 2] += red[3][0];
  }
}

}  // namespace fdtdhip
