// Fused full step: E half step and H half step in ONE pass over memory.
//
// Separate half steps move 24 values per cell and step through HBM (each reads
// three arrays and writes one).  Fused, a cell's step reads E, H, inverse
// permittivity and inverse permeability once and writes the new E and H once:
// 18 values, 25 % less traffic for the same result.
//
// How, without inter-workgroup synchronisation: the step is OUT OF PLACE.  A lane
// owns one cell x.  It reads the old fields (E0, H0), forms the new E of its own
// cell AND of its three forward neighbours x+ex, x+ey, x+ez -- exactly the values
// the curl of the new E at x needs (reference fdtd/grid.py:29-51) -- by evaluating
// the same E update there, then advances H at x and writes E1(x), H1(x) to the
// second pair of buffers.  Nothing a lane reads is written by the launch, so there
// is no ordering requirement between lanes or workgroups.  The re-derived
// neighbour values are bit-identical to what their owner lanes store (same
// operations, same operands), so the result equals two separate half steps.
// The extra loads of the neighbourhood (13 cells of H0, 4 of E0 and eps) are
// L1/L2 hits; HBM sees each array once.  Those 26 loads per lane make the kernel
// L1-bound on large grids (512^3 fp32: 3.2 ms a step against 2.2 ms for two half
// steps), so the plan uses it where launch count decides the step time: small
// grids (large 3-D grids take step_march of yee_march.hpp instead, which marches along x
// over whole z rows).  A first marching variant over 64-cell z segments, planes in registers
// and neighbours through LDS, cut the loads to 8 per cell but its strided 756-byte row
// segments cost more DRAM efficiency than the saved traffic returned (2.33 ms); a
// variant of this kernel that staged the neighbourhood of a 4-row x 64-cell tile in LDS
// (7 coalesced cell loads per lane) also took 2.33 ms with constant materials, against
// 1.66 ms for two half steps: four E updates per cell plus the staging arithmetic cost
// more issue slots than the saved quarter of the traffic is worth.
//
// Side effects stay with the owner lane: only it stores the PML memory variables
// (psi_E to the second psi buffer, because neighbours still read the old one) and
// its new fields.  Additive point sources act on the new E between the two half
// steps (reference fdtd/grid.py:293-295): every lane that derives the new E of a
// source cell adds the source value.
#pragma once

#include "plan.hpp"
#include "yee_generic.hpp"
#include "yee_tiled.hpp"

namespace fdtdhip {

constexpr int MAX_STEP_SOURCES = 8;
constexpr int MAX_STEP_DETECTORS = 4;

template <typename T>
struct StepArgs {
  BulkArgs<T> e;   // E half step: F = E0 (read), G = H0, mat = inverse permittivity, psi -> psi_out
  BulkArgs<T> h;   // H half step: F = H0 (read), mat = inverse permeability (G unused)
  T* E1;           // new fields
  T* H1;
  int nsrc;
  int64_t src_elem[MAX_STEP_SOURCES];  // flat element index (cell*3 + component)
  T src_value[MAX_STEP_SOURCES];       // value added to that element of the new E
  // cell-list detectors read inside the kernel (saves two tiny launches per step):
  // detmap[cell] = (detector << 24 | point index) or -1; null when they are read by
  // their own kernels.  detE/detH: this step's reading row of each detector.
  const int32_t* detmap;
  T* detE[MAX_STEP_DETECTORS];
  T* detH[MAX_STEP_DETECTORS];
};

// The owner lane of a detector cell deposits the step's final E and H there
// (reference fdtd/detectors.py:114-124: E after the sources, H after the H update).
// v: the cell's entry of the detector map (-1: no detector reads the cell), fetched at
// the start of the kernel so that its latency is not added at the end.
template <typename T>
__device__ __forceinline__ void detect_inline(const StepArgs<T>& a, int32_t v, const T e[3],
                                              const T h[3]) {
  if (v < 0) return;
  const int d = v >> 24;
  const int64_t at = (int64_t)(v & 0xFFFFFF) * 3;
  T* oe = a.detE[0];
  T* oh = a.detH[0];
#pragma unroll
  for (int n = 1; n < MAX_STEP_DETECTORS; ++n) {
    oe = n == d ? a.detE[n] : oe;
    oh = n == d ? a.detH[n] : oh;
  }
  oe[at] = e[0]; oe[at + 1] = e[1]; oe[at + 2] = e[2];
  oh[at] = h[0]; oh[at + 1] = h[1]; oh[at + 2] = h[2];
}

// A lane evaluates the E update of up to four cells: its own (index 0, the only one whose
// side effects are kept) and its forward x, y and z neighbours.  The four evaluations
// run side by side -- one walk over the object table and one over the PML faces for all
// of them, the loads of all cells requested before any is used -- because on the small
// grids this kernel serves a step is one short-lived wave per SIMD and its time is the
// length of the chain of dependent memory round trips, not the amount of work.
constexpr int STEP_CELLS = 4;

template <typename T>
struct StepCell {
  int i, j, k;
  int c3;        // flat element offset of the cell (small grids: fits 32 bits)
  bool live;     // false: the neighbour does not exist (or is the cell itself); not evaluated
  Diffs<T> d;
  T cu[3], e[3], m[3], hs[3];
};

// Additive point sources on the new E of the cells (reference fdtd/grid.py:293-295).
template <typename T>
__device__ __forceinline__ void add_sources(const StepArgs<T>& a, StepCell<T> (&c)[STEP_CELLS]) {
#pragma unroll
  for (int n = 0; n < MAX_STEP_SOURCES; ++n) {
    if (n >= a.nsrc) break;
    const int elem = (int)a.src_elem[n];
    bool hit = false;
#pragma unroll
    for (int q = 0; q < STEP_CELLS; ++q) hit |= c[q].live && (unsigned)(elem - c[q].c3) < 3u;
    if (hit) {
#pragma unroll
      for (int q = 0; q < STEP_CELLS; ++q) {
        const int off = c[q].live ? elem - c[q].c3 : -1;
        const T v = a.src_value[n];
#pragma unroll
        for (int x = 0; x < 3; ++x) {   // a choice between values, not between addresses
          const T old = c[q].e[x];
          c[q].e[x] = off == x ? old + v : old;
        }
      }
    }
  }
}

// One PML face acting on the cells in the E half step: the arithmetic of pml_face, with
// the memory variables and profiles of every cell requested before any is used.  The
// normal axis is a run-time (wave-uniform) value here: one copy of the code per face of
// the unrolled face loop instead of three.
template <typename T>
__device__ __forceinline__ void pml_face_cells(const PmlK<T>& P, int nx, T sc,
                                               StepCell<T> (&c)[STEP_CELLS]) {
  const int axis = P.axis;
  const int64_t slab = (int64_t)(axis == 0 ? P.t : nx) * P.sy * P.sz;
  bool in[STEP_CELLS];
  int l[STEP_CELLS];
  int64_t s[STEP_CELLS];
  T b[STEP_CELLS], cc[STEP_CELLS], p0[STEP_CELLS], p1[STEP_CELLS];
#pragma unroll
  for (int q = 0; q < STEP_CELLS; ++q) {
    // (values are copied out of the cell before any choice between them: a choice
    // between two members would become a run-time address into the cell array and
    // push the whole array out of registers)
    const int qi = c[q].i, qj = c[q].j, qk = c[q].k;
    const int coord = axis == 0 ? qi : (axis == 1 ? qj : qk);
    l[q] = coord - P.lo;
    in[q] = c[q].live && l[q] >= 0 && l[q] < P.t;
    b[q] = cc[q] = p0[q] = p1[q] = T(0);
    s[q] = 0;
    if (in[q]) {
      const int li = axis == 0 ? l[q] : qi, lj = axis == 1 ? l[q] : qj,
                lk = axis == 2 ? l[q] : qk;
      s[q] = ((int64_t)li * P.sy + lj) * P.sz + lk;
      b[q] = P.b[l[q]];
      cc[q] = P.c[l[q]];
      p0[q] = P.psi[s[q]];
      p1[q] = P.psi[slab + s[q]];
    }
  }
#pragma unroll
  for (int q = 0; q < STEP_CELLS; ++q) {
    if (in[q]) {
      // u = (axis+1)%3 takes -psi[0], v = (axis+2)%3 takes psi[1]; the differences along
      // the axis of components u and v drive psi[1] and psi[0]
      const T dxy = c[q].d.x_y, dxz = c[q].d.x_z, dyx = c[q].d.y_x, dyz = c[q].d.y_z,
              dzx = c[q].d.z_x, dzy = c[q].d.z_y;
      const T dFu = axis == 0 ? dxy : (axis == 1 ? dyz : dzx);
      const T dFv = axis == 0 ? dxz : (axis == 1 ? dyx : dzy);
      T q0 = p0[q] * b[q], q1 = p1[q] * b[q];
      if (l[q] >= 1) { q0 += dFv * cc[q]; q1 += dFu * cc[q]; }
      if (q == 0) {
        P.psi_out[s[q]] = q0;
        P.psi_out[slab + s[q]] = q1;
      }
      if (P.fused) {
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          const bool is_u = x == (axis + 1) % 3, is_v = x == (axis + 2) % 3;
          const T f = sc * c[q].m[x], old = c[q].e[x];
          const T with_u = old + f * (-q0), with_v = old + f * q1;
          c[q].e[x] = is_u ? with_u : (is_v ? with_v : old);
        }
      }
    }
  }
}

// The E update of the cells: materials, curl, grid update, objects in registration order,
// PML faces in registration order, sources -- per cell the operations of finish_E.
template <typename T, bool HAS_PML, bool HAS_OBJ>
__device__ __forceinline__ void new_E_cells(const StepArgs<T>& a, StepCell<T> (&c)[STEP_CELLS]) {
  const BulkArgs<T>& A = a.e;
  bool hull = false, core = true;
#pragma unroll
  for (int q = 0; q < STEP_CELLS; ++q) {
    if (HAS_OBJ) hull |= c[q].live && in_hull(A, c[q].i, c[q].j, c[q].k);
    if (HAS_PML) core &= !c[q].live || in_core(A, c[q].i, c[q].j, c[q].k);
  }
  const bool obj_near = HAS_OBJ && __any(hull), pml_near = HAS_PML && !__all(core);

  // old E and the inverse permittivity of every cell
#pragma unroll
  for (int q = 0; q < STEP_CELLS; ++q)
    if (c[q].live) load3<T, false>(A.F + c[q].c3, c[q].e[0], c[q].e[1], c[q].e[2]);
  if (A.mat_const) {
    bool covered[STEP_CELLS] = {false, false, false, false};
    if (obj_near)
      for (int o = 0; o < A.nobj; ++o) {
        const ObjK<T> O = load_obj<T>(A, o);
#pragma unroll
        for (int q = 0; q < STEP_CELLS; ++q)
          covered[q] |= c[q].i >= O.x0 && c[q].i < O.x1 && c[q].j >= O.y0 && c[q].j < O.y1 &&
                        c[q].k >= O.z0 && c[q].k < O.z1;
      }
#pragma unroll
    for (int q = 0; q < STEP_CELLS; ++q)
#pragma unroll
      for (int x = 0; x < 3; ++x) c[q].m[x] = covered[q] ? T(0) : A.bg[x];
  } else {
#pragma unroll
    for (int q = 0; q < STEP_CELLS; ++q)
      if (c[q].live) load3<T, false>(A.mat + c[q].c3, c[q].m[0], c[q].m[1], c[q].m[2]);
  }

#pragma unroll
  for (int q = 0; q < STEP_CELLS; ++q) {
    c[q].cu[0] = c[q].d.y_z - c[q].d.z_y;
    c[q].cu[1] = c[q].d.z_x - c[q].d.x_z;
    c[q].cu[2] = c[q].d.x_y - c[q].d.y_x;
#pragma unroll
    for (int x = 0; x < 3; ++x) c[q].e[x] += (A.sc * c[q].m[x]) * c[q].cu[x];
  }

  if (obj_near)
    for (int o = 0; o < A.nobj; ++o) {
      const ObjK<T> O = load_obj<T>(A, o);
#pragma unroll
      for (int q = 0; q < STEP_CELLS; ++q)
        if (c[q].live)
          apply_object<T, false>(A, O, c[q].i, c[q].j, c[q].k, c[q].cu, c[q].e, c[q].hs, q == 0);
    }

  if (pml_near) {
#pragma unroll
    for (int p = 0; p < MAX_PML; ++p) {
      if (p >= A.npml) break;
      const int ax = A.face_axis[p];
      bool hit = false;
#pragma unroll
      for (int q = 0; q < STEP_CELLS; ++q) {
        const int qi = c[q].i, qj = c[q].j, qk = c[q].k;
        const int coord = ax == 0 ? qi : (ax == 1 ? qj : qk);
        hit |= c[q].live && coord >= A.face_lo[p] && coord < A.face_hi[p];
      }
      if (__any(hit)) {
        const PmlK<T> P = A.pml[p];
        pml_face_cells<T>(P, A.nx, A.sc, c);
      }
    }
  }
  add_sources<T>(a, c);
}

template <typename T>
__device__ __forceinline__ void set_cell(StepCell<T>& c, bool live, int i, int j, int k, int c3,
                                         const T hc[3], const T hxm[3], const T hym[3],
                                         const T hzm[3]) {
  c.live = live; c.i = i; c.j = j; c.k = k; c.c3 = c3;
  const bool hasx = i > 0, hasy = j > 0, hasz = k > 0;
  c.d.x_y = hasx ? hc[1] - hxm[1] : T(0);
  c.d.x_z = hasx ? hc[2] - hxm[2] : T(0);
  c.d.y_x = hasy ? hc[0] - hym[0] : T(0);
  c.d.y_z = hasy ? hc[2] - hym[2] : T(0);
  c.d.z_x = hasz ? hc[0] - hzm[0] : T(0);
  c.d.z_y = hasz ? hc[1] - hzm[1] : T(0);
#pragma unroll
  for (int x = 0; x < 3; ++x) { c.hs[x] = hc[x]; c.e[x] = T(0); }
}

template <typename T, bool HAS_PML, bool HAS_OBJ>
__global__ __launch_bounds__(256) void step_fused(StepArgs<T> a) {
  int i, j, k;
  if (!lane_cell<true>(0, a.e.ny, a.e.nz, 0, i, j, k)) return;
  const int nx = a.e.nx, ny = a.e.ny, nz = a.e.nz;
  const int sz = 3, sy = 3 * nz, sx = sy * ny;      // small grids: element offsets fit 32 bits
  const int c3 = i * sx + j * sy + 3 * k;
  const T* __restrict__ H0 = a.e.G + c3;
  const int32_t det = a.detmap ? a.detmap[c3 / 3] : -1;

  // neighbour offsets, clamped to the cell itself where the neighbour does not exist
  const int xm = i > 0 ? -sx : 0, ym = j > 0 ? -sy : 0, zm = k > 0 ? -sz : 0;
  const bool hasxp = i < nx - 1, hasyp = j < ny - 1, haszp = k < nz - 1;
  const int xp = hasxp ? sx : 0, yp = hasyp ? sy : 0, zp = haszp ? sz : 0;

  T hc[3], hxm[3], hym[3], hzm[3], hxp[3], hyp[3], hzp[3];
  T hxpym[3], hxpzm[3], hypxm[3], hypzm[3], hzpxm[3], hzpym[3];
  load3<T, false>(H0, hc[0], hc[1], hc[2]);
  load3<T, false>(H0 + xm, hxm[0], hxm[1], hxm[2]);
  load3<T, false>(H0 + ym, hym[0], hym[1], hym[2]);
  load3<T, false>(H0 + zm, hzm[0], hzm[1], hzm[2]);
  load3<T, false>(H0 + xp, hxp[0], hxp[1], hxp[2]);
  load3<T, false>(H0 + yp, hyp[0], hyp[1], hyp[2]);
  load3<T, false>(H0 + zp, hzp[0], hzp[1], hzp[2]);
  load3<T, false>(H0 + xp + ym, hxpym[0], hxpym[1], hxpym[2]);
  load3<T, false>(H0 + xp + zm, hxpzm[0], hxpzm[1], hxpzm[2]);
  load3<T, false>(H0 + yp + xm, hypxm[0], hypxm[1], hypxm[2]);
  load3<T, false>(H0 + yp + zm, hypzm[0], hypzm[1], hypzm[2]);
  load3<T, false>(H0 + zp + xm, hzpxm[0], hzpxm[1], hzpxm[2]);
  load3<T, false>(H0 + zp + ym, hzpym[0], hzpym[1], hzpym[2]);

  // new E at the cell (index 0: stores psi_E) and at its forward neighbours
  StepCell<T> c[STEP_CELLS];
  set_cell<T>(c[0], true, i, j, k, c3, hc, hxm, hym, hzm);
  set_cell<T>(c[1], hasxp, i + (hasxp ? 1 : 0), j, k, c3 + xp, hxp, hc, hxpym, hxpzm);
  set_cell<T>(c[2], hasyp, i, j + (hasyp ? 1 : 0), k, c3 + yp, hyp, hypxm, hc, hypzm);
  set_cell<T>(c[3], haszp, i, j, k + (haszp ? 1 : 0), c3 + zp, hzp, hzpxm, hzpym, hc);
  new_E_cells<T, HAS_PML, HAS_OBJ>(a, c);
  const T* e = c[0].e;
  store3<T, true>(a.E1 + c3, e[0], e[1], e[2]);

  // H half step at the cell from the new E (forward differences, dropped at the high edges)
  Diffs<T> d;
  d.x_y = hasxp ? c[1].e[1] - e[1] : T(0);
  d.x_z = hasxp ? c[1].e[2] - e[2] : T(0);
  d.y_x = hasyp ? c[2].e[0] - e[0] : T(0);
  d.y_z = hasyp ? c[2].e[2] - e[2] : T(0);
  d.z_x = haszp ? c[3].e[0] - e[0] : T(0);
  d.z_y = haszp ? c[3].e[1] - e[1] : T(0);
  T m[3];
  cell_material<T, true, false>(a.h, c3, i, j, k, m);
  finish_H<T, HAS_PML>(a.h, i, j, k, d, hc, m);
  store3<T, true>(a.H1 + c3, hc[0], hc[1], hc[2]);
  detect_inline<T>(a, det, e, hc);
}

}  // namespace fdtdhip
