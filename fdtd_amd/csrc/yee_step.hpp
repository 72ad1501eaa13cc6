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
// grids.  A marching variant that carried planes in registers and exchanged
// neighbours through LDS cut the loads to 8 per cell but its strided 756-byte row
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
template <typename T>
__device__ __forceinline__ void detect_inline(const StepArgs<T>& a, int64_t c3, const T e[3],
                                              const T h[3]) {
  if (!a.detmap) return;
  const int32_t v = a.detmap[c3 / 3];
  if (v < 0) return;
  const int d = v >> 24;
  const int64_t at = (int64_t)(v & 0xFFFFFF) * 3;
#pragma unroll
  for (int n = 0; n < MAX_STEP_DETECTORS; ++n)
    if (n == d) {
      a.detE[n][at] = e[0]; a.detE[n][at + 1] = e[1]; a.detE[n][at + 2] = e[2];
      a.detH[n][at] = h[0]; a.detH[n][at + 1] = h[1]; a.detH[n][at + 2] = h[2];
    }
}

template <typename T>
__device__ __forceinline__ void add_sources(const StepArgs<T>& a, int64_t c3, T e[3]) {
#pragma unroll
  for (int n = 0; n < MAX_STEP_SOURCES; ++n) {
    if (n >= a.nsrc) break;
    const int64_t off = a.src_elem[n] - c3;
    if (off == 0) e[0] += a.src_value[n];
    else if (off == 1) e[1] += a.src_value[n];
    else if (off == 2) e[2] += a.src_value[n];
  }
}

// New E of cell (i,j,k) whose flat element offset is c3.  hc: H0 at the cell; hxm,
// hym, hzm: H0 at the backward neighbours (any value where that neighbour does not
// exist).
template <typename T, bool HAS_PML, bool HAS_OBJ, bool OWNER>
__device__ __forceinline__ void new_E(const StepArgs<T>& a, int i, int j, int k, int64_t c3,
                                      const T hc[3], const T hxm[3], const T hym[3],
                                      const T hzm[3], T e[3]) {
  Diffs<T> d;
  const bool hasx = i > 0, hasy = j > 0, hasz = k > 0;
  d.x_y = hasx ? hc[1] - hxm[1] : T(0);
  d.x_z = hasx ? hc[2] - hxm[2] : T(0);
  d.y_x = hasy ? hc[0] - hym[0] : T(0);
  d.y_z = hasy ? hc[2] - hym[2] : T(0);
  d.z_x = hasz ? hc[0] - hzm[0] : T(0);
  d.z_y = hasz ? hc[1] - hzm[1] : T(0);
  T m[3];
  load3<T, false>(a.e.F + c3, e[0], e[1], e[2]);
  cell_material<T, false, HAS_OBJ>(a.e, c3, i, j, k, m);
  finish_E<T, HAS_PML, HAS_OBJ>(a.e, i, j, k, d, e, m, hc, OWNER);
  add_sources<T>(a, c3, e);
}

template <typename T, bool HAS_PML, bool HAS_OBJ>
__global__ __launch_bounds__(256) void step_fused(StepArgs<T> a) {
  int i, j, k;
  if (!lane_cell<true>(0, a.e.ny, a.e.nz, 0, i, j, k)) return;
  const int nx = a.e.nx, ny = a.e.ny, nz = a.e.nz;
  const int64_t sz = 3, sy = 3 * (int64_t)nz, sx = sy * ny;
  const int64_t c3 = (int64_t)i * sx + (int64_t)j * sy + 3 * k;
  const T* __restrict__ H0 = a.e.G + c3;

  // neighbour offsets, clamped to the cell itself where the neighbour does not exist
  const int64_t xm = i > 0 ? -sx : 0, ym = j > 0 ? -sy : 0, zm = k > 0 ? -sz : 0;
  const bool hasxp = i < nx - 1, hasyp = j < ny - 1, haszp = k < nz - 1;
  const int64_t xp = hasxp ? sx : 0, yp = hasyp ? sy : 0, zp = haszp ? sz : 0;

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

  // new E at the cell (owner: stores psi_E) and at its forward neighbours
  T e[3], ex[3], ey[3], ez[3];
  new_E<T, HAS_PML, HAS_OBJ, true>(a, i, j, k, c3, hc, hxm, hym, hzm, e);
  new_E<T, HAS_PML, HAS_OBJ, false>(a, i + (hasxp ? 1 : 0), j, k, c3 + xp, hxp, hc, hxpym, hxpzm, ex);
  new_E<T, HAS_PML, HAS_OBJ, false>(a, i, j + (hasyp ? 1 : 0), k, c3 + yp, hyp, hypxm, hc, hypzm, ey);
  new_E<T, HAS_PML, HAS_OBJ, false>(a, i, j, k + (haszp ? 1 : 0), c3 + zp, hzp, hzpxm, hzpym, hc, ez);
  store3<T, true>(a.E1 + c3, e[0], e[1], e[2]);

  // H half step at the cell from the new E (forward differences, dropped at the high edges)
  Diffs<T> d;
  d.x_y = hasxp ? ex[1] - e[1] : T(0);
  d.x_z = hasxp ? ex[2] - e[2] : T(0);
  d.y_x = hasyp ? ey[0] - e[0] : T(0);
  d.y_z = hasyp ? ey[2] - e[2] : T(0);
  d.z_x = haszp ? ez[0] - e[0] : T(0);
  d.z_y = haszp ? ez[1] - e[1] : T(0);
  T m[3];
  cell_material<T, true, false>(a.h, c3, i, j, k, m);
  finish_H<T, HAS_PML>(a.h, i, j, k, d, hc, m);
  store3<T, true>(a.H1 + c3, hc[0], hc[1], hc[2]);
  detect_inline<T>(a, c3, e, hc);
}

}  // namespace fdtdhip
