// Shape-agnostic Yee half-step kernels: one lane per cell, lanes running along
// the flattened (y, z) index so that a wavefront reads 64 adjacent cells
// (64 x 3 components, contiguous).  These kernels carry every feature of the
// half step and are used for any grid shape (2-D grids with nz == 1, thin
// slabs, ragged sizes).  Neighbour values are re-read through L1/L2.
//
// Arithmetic follows the reference expression by expression, in its order, and
// the library is built with -ffp-contract=off, so float64 results equal the
// reference's numpy results (see the citations at each block).
#pragma once

#include "plan.hpp"

namespace fdtdhip {

// One PML face acting on one cell (reference fdtd/boundaries.py:398-476), with the
// normal axis a compile-time constant so that the two transverse components
// u=(AXIS+1)%3, v=(AXIS+2)%3 are static register indices.
//   dFu, dFv : difference along AXIS of components u and v of the driving field
//   phaseE   : E half step: accumulate where the slab-local index is >= 1, F += ...;
//              H half step: accumulate where it is <= t-2, F -= ...
template <typename T, int AXIS>
__device__ __forceinline__ void pml_face(const PmlK<T>& P, int nx, int i, int j, int k, T dFu,
                                         T dFv, bool phaseE, T sc, T f[3], const T m[3],
                                         bool store) {
  constexpr int u = (AXIS + 1) % 3, v = (AXIS + 2) % 3;
  const int coord = AXIS == 0 ? i : (AXIS == 1 ? j : k);
  const int l = coord - P.lo;
  if (l < 0 || l >= P.t) return;
  const int li = AXIS == 0 ? l : i, lj = AXIS == 1 ? l : j, lk = AXIS == 2 ? l : k;
  const int64_t s = ((int64_t)li * P.sy + lj) * P.sz + lk;
  const int64_t slab = (int64_t)(AXIS == 0 ? P.t : nx) * P.sy * P.sz;
  const T b = P.b[l], cc = P.c[l];
  T p0 = P.psi[s] * b, p1 = P.psi[slab + s] * b;
  if (phaseE ? (l >= 1) : (l <= P.t - 2)) { p0 += dFv * cc; p1 += dFu * cc; }
  if (store) {
    P.psi_out[s] = p0;
    P.psi_out[slab + s] = p1;
  }
  if (P.fused) {
    if (phaseE) {
      f[u] += (sc * m[u]) * (-p0);
      f[v] += (sc * m[v]) * p1;
    } else {
      f[u] -= (sc * m[u]) * (-p0);
      f[v] -= (sc * m[v]) * p1;
    }
  }
}

template <typename T>
struct Diffs;

// Inside the box that no PML slab reaches.
template <typename T>
__device__ __forceinline__ bool in_core(const BulkArgs<T>& a, int i, int j, int k) {
  return i >= a.core[0] && i < a.core[1] && j >= a.core[2] && j < a.core[3] && k >= a.core[4] &&
         k < a.core[5];
}

// All PML faces of the plan, in registration order.  The descriptors live in a small
// device table read at wave-uniform addresses (scalar loads), one face at a time, so
// that the loop costs no vector registers.
template <typename T, bool PHASE_E>
__device__ __forceinline__ void pml_all(const BulkArgs<T>& a, int i, int j, int k,
                                        const Diffs<T>& d, T f[3], const T m[3], bool store);

// ---------------------------------------------------------------------------
// E half step: PML memory update (reference fdtd/boundaries.py:422-448),
// curl_H (fdtd/grid.py:54-76), E += sc*inv_eps*curl (grid.py:283), objects
// (fdtd/objects.py:116-126, 204-218, 251-266), PML correction (boundaries.py:398-408).
// ---------------------------------------------------------------------------
// Differences of the driving field around one cell, named d_<axis>_<component>.
template <typename T>
struct Diffs {
  T x_y, x_z, y_x, y_z, z_x, z_y;  // d_<axis>_<component>
};

template <typename T, bool PHASE_E>
__device__ __forceinline__ void pml_all(const BulkArgs<T>& a, int i, int j, int k,
                                        const Diffs<T>& d, T f[3], const T m[3], bool store) {
  if (__all(in_core(a, i, j, k))) return;
  // (a wave that waits for one table entry per face, touched or not, lives about a fifth
  // longer; with a third of all waves near some slab that was 5 % of the 512^3 step)
#pragma unroll
  for (int p = 0; p < MAX_PML; ++p) {
    if (p >= a.npml) break;
    const int ax = a.face_axis[p];
    const int coord = ax == 0 ? i : (ax == 1 ? j : k);
    if (!__any(coord >= a.face_lo[p] && coord < a.face_hi[p])) continue;
    const PmlK<T> P = a.pml[p];
    if (P.axis == 0) pml_face<T, 0>(P, a.nx, i, j, k, d.x_y, d.x_z, PHASE_E, a.sc, f, m, store);
    else if (P.axis == 1) pml_face<T, 1>(P, a.nx, i, j, k, d.y_z, d.y_x, PHASE_E, a.sc, f, m, store);
    else pml_face<T, 2>(P, a.nx, i, j, k, d.z_x, d.z_y, PHASE_E, a.sc, f, m, store);
  }
}

// Everything of the E half step that follows the neighbour differences, for one
// cell: curl, grid update, objects in registration order, PML memory update
// and correction.  `e` holds the cell's E on entry and the new E on return.
// store = false evaluates the cell without side effects (no memory-variable or trace
// writes).
// Inside the box that contains every object.
template <typename T>
__device__ __forceinline__ bool in_hull(const BulkArgs<T>& a, int i, int j, int k) {
  return i >= a.hull[0] && i < a.hull[1] && j >= a.hull[2] && j < a.hull[3] && k >= a.hull[4] &&
         k < a.hull[5];
}

// One object acting on one cell in the E half step (reference fdtd/objects.py:116-126,
// 204-218, 251-266, 328-349); nothing happens outside its box.  cu: curl of H at the cell.
// HAS_MOD: energy-modulated objects (kind 3) may be present.  Their branch (nine
// sigmoids) is compiled only into the kernels that run such plans, so that it does not
// cost the others registers and occupancy.
template <typename T, bool HAS_MOD>
__device__ __forceinline__ void apply_object(const BulkArgs<T>& a, const ObjK<T>& O, int i, int j,
                                             int k, const T cu[3], T e[3], const T hs[3],
                                             bool store) {
  if (i < O.x0 || i >= O.x1 || j < O.y0 || j >= O.y1 || k < O.z0 || k >= O.z1) return;
  const int64_t oc =
      ((int64_t)(i - O.x0) * (O.y1 - O.y0) + (j - O.y0)) * (O.z1 - O.z0) + (k - O.z0);
  if (O.kind == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) e[c] += (a.sc * O.inv[oc * 3 + c]) * cu[c];
  } else if (O.kind == 1) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const T f = O.absorb[oc * 3 + c];
      e[c] *= (T(1) - f) / (T(1) + f);
      e[c] += ((a.sc * O.inv[oc * 3 + c]) * cu[c]) / (T(1) + f);
    }
  } else if (HAS_MOD && O.kind == 3) {
    // tensor entries scaled by sigmoid(w_E*sum(E^2) + w_H*sum(H^2) + bias), the
    // energies taken from the fields as they are at this point of the half step
    // (reference fdtd/objects.py:328-349)
    const T eE = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
    const T eH = hs[0] * hs[0] + hs[1] * hs[1] + hs[2] * hs[2];
    if (store && a.trace_step >= 0) {
      const int64_t box = (int64_t)(O.x1 - O.x0) * (O.y1 - O.y0) * (O.z1 - O.z0);
      if (O.trace) {
        T* tr = O.trace + (a.trace_step * box + oc) * 3;
        tr[0] = cu[0]; tr[1] = cu[1]; tr[2] = cu[2];
      }
      if (O.etrace) {
        T* et = O.etrace + (a.trace_step * box + oc) * 2;
        et[0] = eE; et[1] = eH;
      }
    }
    T add[3];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      const T* M = O.inv + oc * 9 + rr * 3;
      T s = T(0);
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const T* w = O.modw + (rr * 3 + cc) * 3;
        const T mod = T(1) / (T(1) + exp(-(w[0] * eE + w[1] * eH + w[2])));
        s += (M[cc] * mod) * cu[cc];
      }
      add[rr] = a.sc * s;
    }
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) e[rr] += add[rr];
  } else {
    if (store && O.trace && a.trace_step >= 0) {  // differentiable run: keep the curl
      const int64_t box =
          (int64_t)(O.x1 - O.x0) * (O.y1 - O.y0) * (O.z1 - O.z0);
      T* tr = O.trace + (a.trace_step * box + oc) * 3;
      tr[0] = cu[0]; tr[1] = cu[1]; tr[2] = cu[2];
    }
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      const T* M = O.inv + oc * 9 + rr * 3;
      T s = M[0] * cu[0];
      s += M[1] * cu[1];
      s += M[2] * cu[2];
      e[rr] += a.sc * s;
    }
  }
}

template <typename T, bool HAS_PML, bool HAS_OBJ, bool HAS_MOD = false>
__device__ __forceinline__ void finish_E(const BulkArgs<T>& a, int i, int j, int k,
                                         const Diffs<T>& d, T e[3], const T m[3],
                                         const T hs[3], bool store = true) {
  T cu[3];
  cu[0] = d.y_z - d.z_y;
  cu[1] = d.z_x - d.x_z;
  cu[2] = d.x_y - d.y_x;
#pragma unroll
  for (int c = 0; c < 3; ++c) e[c] += (a.sc * m[c]) * cu[c];

  if (HAS_OBJ && __any(in_hull(a, i, j, k))) {
    for (int o = 0; o < a.nobj; ++o) {
      const ObjK<T> O = load_obj<T>(a, o);
      apply_object<T, HAS_MOD>(a, O, i, j, k, cu, e, hs, store);
    }
  }

  if (HAS_PML) pml_all<T, true>(a, i, j, k, d, e, m, store);
}

// Same for the H half step (no objects: reference fdtd/objects.py:128-134).
template <typename T, bool HAS_PML>
__device__ __forceinline__ void finish_H(const BulkArgs<T>& a, int i, int j, int k,
                                         const Diffs<T>& d, T h[3], const T m[3],
                                         bool store = true) {
  T cu[3];
  cu[0] = d.y_z - d.z_y;
  cu[1] = d.z_x - d.x_z;
  cu[2] = d.x_y - d.y_x;
#pragma unroll
  for (int c = 0; c < 3; ++c) h[c] -= (a.sc * m[c]) * cu[c];

  if (HAS_PML) pml_all<T, false>(a, i, j, k, d, h, m, store);
}

// Inverse material of one cell.  Grids are usually homogeneous outside their objects
// (reference fdtd/grid.py:135-151 fills both arrays with one value; fdtd/objects.py:90
// zeroes the inverse permittivity under an object), and a scan of the array
// (material_scan below) can prove it: the value is then a kernel argument and a
// quarter of the half step's memory traffic disappears.  The result is the same
// either way -- the constant is the value the array holds.
template <typename T, bool NT, bool HAS_OBJ>
__device__ __forceinline__ void cell_material(const BulkArgs<T>& a, int64_t c3, int i, int j,
                                              int k, T m[3]) {
  if (a.mat_const) {
    bool covered = false;
    if (HAS_OBJ && __any(in_hull(a, i, j, k)))
      for (int o = 0; o < a.nobj; ++o) {
        const ObjK<T> O = load_obj<T>(a, o);
        covered |= i >= O.x0 && i < O.x1 && j >= O.y0 && j < O.y1 && k >= O.z0 && k < O.z1;
      }
#pragma unroll
    for (int c = 0; c < 3; ++c) m[c] = covered ? T(0) : a.bg[c];
  } else if (NT) {
    m[0] = __builtin_nontemporal_load(a.mat + c3);
    m[1] = __builtin_nontemporal_load(a.mat + c3 + 1);
    m[2] = __builtin_nontemporal_load(a.mat + c3 + 2);
  } else {
    m[0] = a.mat[c3]; m[1] = a.mat[c3 + 1]; m[2] = a.mat[c3 + 2];
  }
}

__device__ __forceinline__ bool same_bits(float x, float y) {
  return __float_as_uint(x) == __float_as_uint(y);
}
__device__ __forceinline__ bool same_bits(double x, double y) {
  return __double_as_longlong(x) == __double_as_longlong(y);
}

// Sets *differs when some cell of `mat` is not, bit for bit, bg outside the nobj
// boxes and +0 inside them.
template <typename T>
__global__ __launch_bounds__(256) void material_scan(const T* __restrict__ mat, int64_t ncell,
                                                     int ny, int nz, const ObjK<T>* obj,
                                                     int nobj, T bg0, T bg1, T bg2,
                                                     int* differs) {
  const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= ncell) return;
  bool covered = false;
  if (nobj) {
    const int k = (int)(cell % nz), j = (int)((cell / nz) % ny), i = (int)(cell / ((int64_t)nz * ny));
    for (int o = 0; o < nobj; ++o) {
      const ObjK<T> O = obj[o];
      covered |= i >= O.x0 && i < O.x1 && j >= O.y0 && j < O.y1 && k >= O.z0 && k < O.z1;
    }
  }
  const T* m = mat + cell * 3;
  const bool ok = same_bits(m[0], covered ? T(0) : bg0) && same_bits(m[1], covered ? T(0) : bg1) &&
                  same_bits(m[2], covered ? T(0) : bg2);
  if (!ok) *differs = 1;
}

// Block index -> block of work.  With BAND, the 8 XCDs (blocks are dealt to them
// round robin, so blocks b and b+8 share an L2) each take a contiguous band of
// rows of every x plane: the y- and x-neighbours of a cell are then fetched by
// the same XCD and hit its L2 instead of crossing the fabric again.  Purely a
// speed matter: any mapping gives the same results.
template <bool BAND>
__device__ __forceinline__ int64_t work_block(int64_t blocks_per_plane) {
  const int64_t b = blockIdx.x;
  if (!BAND) return b;
  const int64_t per = blocks_per_plane >> 3;
  const int64_t xcd = b & 7, s = b >> 3;
  return (s / per) * blocks_per_plane + xcd * per + (s % per);
}

template <typename T, bool HAS_PML, bool HAS_OBJ, bool BAND, bool HAS_MOD = false>
__global__ __launch_bounds__(256) void bulk_E_generic(BulkArgs<T> a) {
  const int64_t plane = (int64_t)a.ny * a.nz;
  const int64_t ncell = (int64_t)(a.x1 - a.x0) * plane;
  const int64_t tid = work_block<BAND>(plane >> 8) * blockDim.x + threadIdx.x;
  if (tid >= ncell) return;
  const int i = a.x0 + (int)(tid / plane);
  const int64_t r = tid % plane;
  const int j = (int)(r / a.nz);
  const int k = (int)(r % a.nz);
  const int64_t sz = 3, sy = 3 * (int64_t)a.nz, sx = 3 * plane;
  const int64_t c3 = ((int64_t)i * plane + r) * 3;

  const T* __restrict__ G = a.G;
  const T hx = G[c3], hy = G[c3 + 1], hz = G[c3 + 2];
  Diffs<T> d = {0, 0, 0, 0, 0, 0};
  if (j > 0) { d.y_z = hz - G[c3 - sy + 2]; d.y_x = hx - G[c3 - sy]; }
  if (k > 0) { d.z_y = hy - G[c3 - sz + 1]; d.z_x = hx - G[c3 - sz]; }
  if (i > 0) { d.x_z = hz - G[c3 - sx + 2]; d.x_y = hy - G[c3 - sx + 1]; }
  else if (a.ghost) { d.x_z = hz - a.ghost[r * 2 + 1]; d.x_y = hy - a.ghost[r * 2]; }

  T e[3] = {a.F[c3], a.F[c3 + 1], a.F[c3 + 2]};
  T m[3];
  cell_material<T, false, HAS_OBJ>(a, c3, i, j, k, m);
  const T hs[3] = {hx, hy, hz};
  finish_E<T, HAS_PML, HAS_OBJ, HAS_MOD>(a, i, j, k, d, e, m, hs);
  a.F[c3] = e[0];
  a.F[c3 + 1] = e[1];
  a.F[c3 + 2] = e[2];
}

// ---------------------------------------------------------------------------
// H half step: PML memory update (reference fdtd/boundaries.py:450-476),
// curl_E (fdtd/grid.py:29-51), H -= sc*inv_mu*curl (grid.py:309), PML
// correction (boundaries.py:410-420).  Objects do nothing in this half step
// (fdtd/objects.py:128-134, 220-226, 268-274).
// ---------------------------------------------------------------------------
template <typename T, bool HAS_PML, bool BAND>
__global__ __launch_bounds__(256) void bulk_H_generic(BulkArgs<T> a) {
  const int64_t plane = (int64_t)a.ny * a.nz;
  const int64_t ncell = (int64_t)(a.x1 - a.x0) * plane;
  const int64_t tid = work_block<BAND>(plane >> 8) * blockDim.x + threadIdx.x;
  if (tid >= ncell) return;
  const int i = a.x0 + (int)(tid / plane);
  const int64_t r = tid % plane;
  const int j = (int)(r / a.nz);
  const int k = (int)(r % a.nz);
  const int64_t sz = 3, sy = 3 * (int64_t)a.nz, sx = 3 * plane;
  const int64_t c3 = ((int64_t)i * plane + r) * 3;

  const T* __restrict__ G = a.G;
  const T ex = G[c3], ey = G[c3 + 1], ez = G[c3 + 2];
  Diffs<T> d = {0, 0, 0, 0, 0, 0};
  if (j < a.ny - 1) { d.y_z = G[c3 + sy + 2] - ez; d.y_x = G[c3 + sy] - ex; }
  if (k < a.nz - 1) { d.z_y = G[c3 + sz + 1] - ey; d.z_x = G[c3 + sz] - ex; }
  if (i < a.nx - 1) { d.x_z = G[c3 + sx + 2] - ez; d.x_y = G[c3 + sx + 1] - ey; }
  else if (a.ghost) { d.x_z = a.ghost[r * 2 + 1] - ez; d.x_y = a.ghost[r * 2] - ey; }

  T h[3] = {a.F[c3], a.F[c3 + 1], a.F[c3 + 2]};
  T m[3];
  cell_material<T, false, false>(a, c3, i, j, k, m);
  finish_H<T, HAS_PML>(a, i, j, k, d, h, m);
  a.F[c3] = h[0];
  a.F[c3 + 1] = h[1];
  a.F[c3 + 2] = h[2];
}

// ---------------------------------------------------------------------------
// Pieces that follow the bulk pass in boundary-list order.
// ---------------------------------------------------------------------------

// Field correction of a PML that was registered after a periodic boundary
// (reference fdtd/boundaries.py:398-420); its memory variables were already
// advanced by the bulk kernel.  sign = +1 for E, -1 for H.
template <typename T>
__global__ void pml_apply_slab(T* F, const T* mat, PmlK<T> P, int nx, int ny, int nz, T sc,
                               T sign) {
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
  const T p0 = P.psi[s], p1 = P.psi[slab + s];
  F[c3 + u] += sign * ((sc * mat[c3 + u]) * (-p0));
  F[c3 + v] += sign * ((sc * mat[c3 + v]) * p1);
}

// Periodic copy of one plane (reference fdtd/boundaries.py:173-208):
// F[dst along axis] = F[src along axis].
template <typename T>
__global__ void periodic_copy(T* F, int nx, int ny, int nz, int axis, int dst, int src) {
  const int n1 = axis == 0 ? ny : nx;
  const int n2 = axis == 2 ? ny : nz;
  const int64_t total = (int64_t)n1 * n2 * 3;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c = (int)(t % 3);
  const int b = (int)((t / 3) % n2);
  const int a = (int)(t / (3 * (int64_t)n2));
  int64_t d, s;
  if (axis == 0) {
    d = (((int64_t)dst * ny + a) * nz + b) * 3 + c;
    s = (((int64_t)src * ny + a) * nz + b) * 3 + c;
  } else if (axis == 1) {
    d = (((int64_t)a * ny + dst) * nz + b) * 3 + c;
    s = (((int64_t)a * ny + src) * nz + b) * 3 + c;
  } else {
    d = (((int64_t)a * ny + b) * nz + dst) * 3 + c;
    s = (((int64_t)a * ny + b) * nz + src) * 3 + c;
  }
  F[d] = F[s];
}

// Additive point list source (reference fdtd/sources.py:111, 336-337):
// F[elem[p]] += profile[p] * w, evaluated in double and rounded once to T.
// `serial` keeps the reference's sequential order when a cell is listed twice.
template <typename T>
__global__ void source_add_points(T* F, const int64_t* elem, const double* profile, double w,
                                  int64_t n, int serial, int64_t lo, int64_t hi) {
  // only elements in [lo, hi) (a range of x planes) are touched
  if (serial) {
    if (blockIdx.x == 0 && threadIdx.x == 0)
      for (int64_t p = 0; p < n; ++p)
        if (elem[p] >= lo && elem[p] < hi) F[elem[p]] += (T)(profile[p] * w);
    return;
  }
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n && elem[p] >= lo && elem[p] < hi) F[elem[p]] += (T)(profile[p] * w);
}

// Hard plane source (reference fdtd/sources.py:519, 525): F[box, comp] = value.
template <typename T>
__global__ void source_set_box(T* F, int ny, int nz, int x0, int x1, int y0, int y1, int z0,
                               int z1, int comp, T value) {
  const int bz = z1 - z0, by = y1 - y0;
  const int64_t total = (int64_t)(x1 - x0) * by * bz;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int k = z0 + (int)(t % bz);
  const int j = y0 + (int)((t / bz) % by);
  const int i = x0 + (int)(t / ((int64_t)bz * by));
  F[(((int64_t)i * ny + j) * nz + k) * 3 + comp] = value;
}

// Additive source over a box whose strength depends on the local field energies
// (reference fdtd/sources.py:581-608, CorticalColumnPlaneSource.update_E):
//   E_comp(cell) += sum_c osc[c] * P[c][column] * sigmoid(w[c][0]*sum(E^2) + w[c][1]*sum(H^2) + w[c][2])
// with `column` the (x, y) position inside the box; the caller folds the activation
// image, the normalised direction kernel's footprint and the amplitude into P and
// evaluates the oscillators of the step into osc.  With etrace the two energies are kept
// for the adjoint.
template <typename T>
__global__ void source_modulated(T* E, const T* H, int ny, int nz, int x0, int x1, int y0, int y1,
                                 int z0, int z1, int comp, const T* __restrict__ P,
                                 const T* __restrict__ osc, const T* __restrict__ wb, int channels,
                                 T* etrace) {
  const int bz = z1 - z0, by = y1 - y0;
  const int64_t cols = (int64_t)(x1 - x0) * by, total = cols * bz;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t col = t / bz;
  const int k = z0 + (int)(t % bz);
  const int j = y0 + (int)(col % by);
  const int i = x0 + (int)(col / by);
  const int64_t c3 = (((int64_t)i * ny + j) * nz + k) * 3;
  const T eE = E[c3] * E[c3] + E[c3 + 1] * E[c3 + 1] + E[c3 + 2] * E[c3 + 2];
  const T eH = H[c3] * H[c3] + H[c3 + 1] * H[c3 + 1] + H[c3 + 2] * H[c3 + 2];
  if (etrace) { etrace[t * 2] = eE; etrace[t * 2 + 1] = eH; }
  T sum = T(0);
  for (int c = 0; c < channels; ++c) {
    const T mod = T(1) / (T(1) + exp(-(wb[c * 3] * eE + wb[c * 3 + 1] * eH + wb[c * 3 + 2])));
    sum += (osc[c] * P[c * cols + col]) * mod;
  }
  E[c3 + comp] += sum;
}

// Voltage source with a series resistance (reference fdtd/sources.py:708-737): one lane.
//   v_in = waveform[q] (0 past its end); v_out = v_in + Z * I_prev when Z > 0, I_prev the
//   loop current read after the previous step (0 at q = 0); E[elem] += v_out / spacing.
// Both voltages are recorded.
template <typename T>
__global__ void source_soft_point(T* E, int64_t elem, const double* wave, int64_t nwave,
                                  int64_t q, double impedance, double spacing, const T* cur,
                                  T* vin_out, T* vout_out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const double vin = q < nwave ? wave[q] : 0.0;
  const double current = (q > 0 && cur) ? (double)cur[0] : 0.0;
  const double vout = impedance > 0 ? vin + impedance * current : vin;
  E[elem] += (T)(vout / spacing);
  if (vin_out) vin_out[0] = (T)vin;
  if (vout_out) vout_out[0] = (T)vout;
}

// Cell-list detectors (reference fdtd/detectors.py:117, 123, 250, 263): out[t] =
// F[cell[t / 3]][t % 3].  Several detectors share a launch: detector d owns the threads
// [first[d], first[d+1]) and reads its own cell list into its own output row.
constexpr int MAX_DET_BATCH = 8;
template <typename T>
struct DetBatch {
  const int64_t* cell[MAX_DET_BATCH];
  T* out[MAX_DET_BATCH];
  int64_t first[MAX_DET_BATCH + 1];  // in values (3 per cell)
  int count;
};
template <typename T>
__global__ void detect_cells_batch(const T* F, DetBatch<T> b) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= b.first[b.count]) return;
#pragma unroll
  for (int d = 0; d < MAX_DET_BATCH; ++d)
    if (d < b.count && t >= b.first[d] && t < b.first[d + 1]) {
      const int64_t u = t - b.first[d];
      b.out[d][u] = F[b.cell[d][u / 3] * 3 + (u % 3)];
    }
}

// The same, leaving out the cells inside `inner` and outside `hole`: a two-step kernel has
// written their readings itself (yee_march2.hpp).
struct DetRegion { int ny, nz; int inner[6]; int hole[6]; };
template <typename T>
__global__ void detect_cells_batch_outside(const T* F, DetBatch<T> b, DetRegion g) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= b.first[b.count]) return;
#pragma unroll
  for (int d = 0; d < MAX_DET_BATCH; ++d)
    if (d < b.count && t >= b.first[d] && t < b.first[d + 1]) {
      const int64_t u = t - b.first[d];
      const int64_t cell = b.cell[d][u / 3], plane = (int64_t)g.ny * g.nz;
      const int i = (int)(cell / plane), j = (int)((cell % plane) / g.nz), k = (int)(cell % g.nz);
      const bool in = i >= g.inner[0] && i < g.inner[1] && j >= g.inner[2] && j < g.inner[3] &&
                      k >= g.inner[4] && k < g.inner[5];
      const bool holed = i >= g.hole[0] && i < g.hole[1] && j >= g.hole[2] && j < g.hole[3] &&
                         k >= g.hole[4] && k < g.hole[5];
      if (!in || holed) b.out[d][u] = F[cell * 3 + (u % 3)];
    }
}

// Loop-current detector (reference fdtd/detectors.py:443-462), including the
// carry of the first layer's Hy term into the second layer (line 456) and
// Python's wrap-around for index -1.
template <typename T>
__global__ void detect_current(const T* H, const T* below, const int32_t* xyz, T* out, int64_t n,
                               int nx, int ny, int nz, T spacing) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int px = xyz[3 * t], py = xyz[3 * t + 1], pz = xyz[3 * t + 2];
  const int xm = px == 0 ? nx - 1 : px - 1;
  const int ym = py == 0 ? ny - 1 : py - 1;
  const int zm = pz == 0 ? nz - 1 : pz - 1;
  auto at = [&](int i, int j, int k, int c) {
    return H[(((int64_t)i * ny + j) * nz + k) * 3 + c];
  };
  // plane px - 1: on an x-slab the plane under plane 0 belongs to another rank (`below`)
  auto under = [&](int j, int k, int c) {
    return px == 0 && below ? below[((int64_t)j * nz + k) * 3 + c] : at(xm, j, k, c);
  };
  T v1 = (at(px, ym, pz, 0) - at(px, py, pz, 0)) * spacing;
  T v2 = (at(px, py, pz, 1) - under(py, pz, 1)) * spacing;
  const T cur1 = v1 + v2;
  v1 = (at(px, ym, zm, 0) - at(px, py, zm, 0)) * spacing;
  v2 += (at(px, py, zm, 1) - under(py, zm, 1)) * spacing;
  const T cur2 = v1 + v2;
  out[t] = (cur1 + cur2) / T(2);
}

// Energy densities sum(E^2), sum(H^2) over the components on one grid plane, in the
// orientation the reference's plots use (reference fdtd/visualization.py:119-146):
// axis 0 -> (ny, nz), axis 1 -> (nz, nx) (transposed), axis 2 -> (nx, ny).
template <typename T>
__global__ void plane_energy(const T* E, const T* H, int nx, int ny, int nz, int axis, int index,
                             T* outE, T* outH) {
  const int rows = axis == 0 ? ny : (axis == 1 ? nz : nx);
  const int cols = axis == 0 ? nz : (axis == 1 ? nx : ny);
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)rows * cols) return;
  const int r = (int)(t / cols), c = (int)(t % cols);
  const int i = axis == 0 ? index : (axis == 1 ? c : r);
  const int j = axis == 0 ? r : (axis == 1 ? index : c);
  const int k = axis == 0 ? c : (axis == 1 ? r : index);
  const int64_t c3 = (((int64_t)i * ny + j) * nz + k) * 3;
  outE[t] = E[c3] * E[c3] + E[c3 + 1] * E[c3 + 1] + E[c3 + 2] * E[c3 + 2];
  outH[t] = H[c3] * H[c3] + H[c3 + 1] * H[c3 + 1] + H[c3 + 2] * H[c3 + 2];
}

// Stand-alone curls (reference fdtd/grid.py:29-76).
template <typename T, bool FORWARD>
__global__ void curl_only(const T* G, T* out, int nx, int ny, int nz) {
  const int64_t plane = (int64_t)ny * nz;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= plane * nx) return;
  const int i = (int)(tid / plane);
  const int64_t r = tid % plane;
  const int j = (int)(r / nz), k = (int)(r % nz);
  const int64_t sz = 3, sy = 3 * (int64_t)nz, sx = 3 * plane, c3 = tid * 3;
  const T gx = G[c3], gy = G[c3 + 1], gz = G[c3 + 2];
  T d_y_z = 0, d_y_x = 0, d_z_y = 0, d_z_x = 0, d_x_z = 0, d_x_y = 0;
  if (FORWARD) {
    if (j < ny - 1) { d_y_z = G[c3 + sy + 2] - gz; d_y_x = G[c3 + sy] - gx; }
    if (k < nz - 1) { d_z_y = G[c3 + sz + 1] - gy; d_z_x = G[c3 + sz] - gx; }
    if (i < nx - 1) { d_x_z = G[c3 + sx + 2] - gz; d_x_y = G[c3 + sx + 1] - gy; }
  } else {
    if (j > 0) { d_y_z = gz - G[c3 - sy + 2]; d_y_x = gx - G[c3 - sy]; }
    if (k > 0) { d_z_y = gy - G[c3 - sz + 1]; d_z_x = gx - G[c3 - sz]; }
    if (i > 0) { d_x_z = gz - G[c3 - sx + 2]; d_x_y = gy - G[c3 - sx + 1]; }
  }
  out[c3] = d_y_z - d_z_y;
  out[c3 + 1] = d_z_x - d_x_z;
  out[c3 + 2] = d_x_y - d_y_x;
}

}  // namespace fdtdhip
