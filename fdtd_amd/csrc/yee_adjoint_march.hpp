// Whole reverse step in one pass for large 3-D grids: the adjoint counterpart of the marching
// step (yee_march.hpp).
//
// Where a cell and its neighbours are "plain" -- no PML slab, outside the hull of the objects,
// constant materials (adj_box_plain in yee_adjoint.hpp) -- the adjoint of a time step
// (reference examples/learnable_focus.py:117-128 differentiates through fdtd/grid.py:267-273)
// has the shape of the step itself:
//
//   undo the H half step:   lamE'(x) = lamE(x) + D-(gH)(x),   gH = -(sc mu) lamH
//                           (differences towards x - e: what the E half step reads of H)
//   undo the E half step:   lamH'(x) = lamH(x) + D+(gE)(x),   gE =  (sc eps) lamE'
//                           (differences towards x + e: what the H half step reads of E)
//
// so the reverse step marches along x exactly as step_march_box does: a workgroup owns TY rows
// of whole z rows and walks a chunk of planes; a lane keeps lamH and the new lamE of the
// previous plane in registers; the z neighbour's new lamE comes by lane shift (across waves
// through LDS); the row below a tile, the plane past a chunk are evaluated redundantly, and
// where those lie outside the box the new lamE is READ from the second buffer, where
// adj_shell_H (the cells next to PML slabs and objects, below) has put it before this launch.  Out of place: (lamE, lamH) -> (lamE', lamH'), one read and one
// write of each adjoint field per reverse step instead of 21 values per cell and half step.
//
// The sums are those of adj_gather_direct, operand for operand: bitwise equal to the two-pass
// reverse step.
#pragma once

#include "plan.hpp"
#include "yee_adjoint.hpp"

namespace fdtdhip {

constexpr int ADJ_MARCH_MAX_WAVES = 16;   // z rows of up to 1024 cells

template <typename T>
struct AdjMarchArgs {
  const T* lamE0;   // adjoint of E after the step being undone
  const T* lamH0;   // adjoint of H after it
  T* lamE1;         // adjoint of E before the step; cells outside the box / inside the hole:
                    // read only (adj_shell_H wrote them)
  T* lamH1;         // adjoint of H before the step (inside the hole: scratch, overwritten by
                    // adj_shell_E after this launch)
  int nx, ny, nz;
  T sc;
  T bgE[3], bgH[3];   // the constant inverse permittivity / permeability
  int open_lo, open_hi;  // x-slab faces that continue on another rank
  int lx;           // planes per chunk
  int band;         // tiles per XCD (grid x = 8 * band), or 0: tile = blockIdx.x
  int chunk0;       // first chunk of planes of this launch
  int box[6];       // x0, x1, y0, y1, z0, z1 of the cells to take the whole reverse step
  int hole[6];      // a box inside `box` left to adj_shell_H / adj_shell_E (objects and the
                    // cells around them), or all zero
};

// The six difference adjoints of a plain cell from the field adjoint at it, as
// adj_plain_G forms them; `kk` = sc * material.  mx / my / mz: the cell sits on the edge
// where the forward pass dropped its x / y / z differences.
template <typename T, bool PHASE_E>
__device__ __forceinline__ void adj_march_G(const T kk[3], const T lam[3], bool mx, bool my, bool mz,
                                            T G[6]) {
  T g[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const T v = kk[c] * lam[c];
    g[c] = PHASE_E ? v : -v;
  }
  G[WY_Z] = g[0]; G[WZ_Y] = -g[0];
  G[WZ_X] = g[1]; G[WX_Z] = -g[1];
  G[WX_Y] = g[2]; G[WY_X] = -g[2];
  if (mx) G[WX_Y] = G[WX_Z] = 0;
  if (my) G[WY_X] = G[WY_Z] = 0;
  if (mz) G[WZ_X] = G[WZ_Y] = 0;
}

template <typename T, int TY, bool HAS_HOLE, int MW = ADJ_MARCH_MAX_WAVES>
__global__ __launch_bounds__(64 * MW) void adj_march_box(AdjMarchArgs<T> a) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = w * 64 + lane;
  const int nx = a.nx, ny = a.ny, nz = a.nz;
  const int bx0 = a.box[0], bx1 = a.box[1], by0 = a.box[2], by1 = a.box[3], bz0 = a.box[4],
            bz1 = a.box[5];
  // lanes that advance a cell; the lane after the last of them fetches its cell's new lamE
  const bool zin = k >= bz0 && k < bz1;
  const bool zedge = k == bz1 && bz1 < nz;
  const bool zhole = HAS_HOLE && k >= a.hole[4] && k < a.hole[5];
  const int tile = a.band > 0 ? (int)(blockIdx.x & 7) * a.band + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int j0 = by0 + tile * TY;
  if (j0 >= by1) return;
  const int i0 = bx0 + ((int)blockIdx.y + a.chunk0) * a.lx, i1 = min(i0 + a.lx, bx1);
  const int64_t sy = 3 * (int64_t)nz, sx = sy * ny;
  const T* __restrict__ H0 = a.lamH0;
  const T* __restrict__ E0 = a.lamE0;
  const T* E1r = a.lamE1;    // cells of adj_shell_H only: not written by this launch
  T kE[3], kH[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { kE[c] = a.sc * a.bgE[c]; kH[c] = a.sc * a.bgH[c]; }
  // new lamE (components x, y) of each wave's first lane, for the last lane of the wave before
  __shared__ T xch[2][ADJ_MARCH_MAX_WAVES + 1][TY][2];
  T Hp[TY + 1][3], Ep[TY + 1][3];
#pragma unroll
  for (int r = 0; r <= TY; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) { Hp[r][c] = T(0); Ep[r][c] = T(0); }
  if (i0 > 0 && zin) {
#pragma unroll
    for (int r = 0; r <= TY; ++r) {
      const int j = j0 + r;
      if (j < by1) {
        const T* p = H0 + (int64_t)(i0 - 1) * sx + (int64_t)j * sy + 3 * k;
        Hp[r][0] = p[0]; Hp[r][1] = p[1]; Hp[r][2] = p[2];
      }
    }
  }
  const bool hasz = k > 0, hasz1 = k < nz - 1;
  const T zero = T(0);
  for (int i = i0; i <= i1; ++i) {
    T Hc[TY + 1][3], En[TY + 1][3], zx[TY + 1], zy[TY + 1], ux = T(0), uz = T(0);
#pragma unroll
    for (int r = 0; r <= TY; ++r) {
      zx[r] = zy[r] = T(0);
#pragma unroll
      for (int c = 0; c < 3; ++c) { Hc[r][c] = T(0); En[r][c] = T(0); }
    }
    const bool plane = i < nx;
    const bool pin = i < bx1;      // a plane of the box (else the plane past it, if any)
    bool holed[TY + 1];            // rows of this plane that cross the hole
#pragma unroll
    for (int r = 0; r <= TY; ++r)
      holed[r] = HAS_HOLE && i >= a.hole[0] && i < a.hole[1] && j0 + r >= a.hole[2] && j0 + r < a.hole[3];
    // Every load of the plane in ONE batch: no test per row around the loads (with one, the
    // compiler closes each row's block with copies of what it loaded and waits for them, a
    // round trip per row).  Rows and planes that do not exist are clamped to ones that do and
    // what they deliver is never used: every use below is masked.  A cell outside the box or
    // inside the hole delivers its new lamE from the second buffer, where adj_shell_H has put
    // it, chosen by address.
    if (zin || zedge) {
      const int64_t pbase = (int64_t)min(i, nx - 1) * sx + 3 * k;
#pragma unroll
      for (int r = 0; r <= TY; ++r) {
        const int64_t c3 = pbase + (int64_t)min(j0 + r, ny - 1) * sy;
        const bool mine = pin && zin && j0 + r < by1 && !(zhole && holed[r]);
        const T* e = mine ? E0 : E1r;
        Hc[r][0] = H0[c3]; Hc[r][1] = H0[c3 + 1]; Hc[r][2] = H0[c3 + 2];
        En[r][0] = e[c3]; En[r][1] = e[c3 + 1]; En[r][2] = e[c3 + 2];
        const T* z = H0 + (hasz ? c3 - 3 : c3);
        zx[r] = z[0]; zy[r] = z[1];
      }
      const T* u = H0 + pbase + (int64_t)max(j0 - 1, 0) * sy;
      ux = u[0]; uz = u[2];
    }
    if (plane && (zin || zedge)) {
      const int64_t pbase = (int64_t)i * sx + 3 * k;
      const bool ox = i > 0;
      if (pin && zin) {
#pragma unroll
        for (int r = 0; r <= TY; ++r) {
          const int j = j0 + r;
          if (j >= by1) continue;
          const bool keep = zhole && holed[r];   // a cell of the hole: lamE1 as loaded
          const int64_t c3 = pbase + (int64_t)j * sy;
          const bool oy = j > 0;
          // undo the H half step at (i, j, k): gather from the cell and its three low neighbours
          const T vx[3] = {zero, Hp[r][1], Hp[r][2]};
          const T vy[3] = {r == 0 ? ux : Hc[r == 0 ? 0 : r - 1][0], zero, r == 0 ? uz : Hc[r == 0 ? 0 : r - 1][2]};
          const T vz[3] = {zx[r], zy[r], zero};
          T g[6], gx[6], gy[6], gz[6];
          adj_march_G<T, false>(kH, Hc[r], i == nx - 1 && !a.open_hi, j == ny - 1, k == nz - 1, g);
          adj_march_G<T, false>(kH, vx, false, false, false, gx);
          adj_march_G<T, false>(kH, vy, false, false, false, gy);
          adj_march_G<T, false>(kH, vz, false, false, false, gz);
          T add[3];
          add[0] = ((oy ? gy[WY_X] : zero) - g[WY_X]) + ((hasz ? gz[WZ_X] : zero) - g[WZ_X]);
          add[1] = ((ox ? gx[WX_Y] : zero) - g[WX_Y]) + ((hasz ? gz[WZ_Y] : zero) - g[WZ_Y]);
          add[2] = ((ox ? gx[WX_Z] : zero) - g[WX_Z]) + ((oy ? gy[WY_Z] : zero) - g[WY_Z]);
#pragma unroll
          for (int c = 0; c < 3; ++c) En[r][c] = keep ? En[r][c] : En[r][c] + add[c];
          // (a cell of the hole stores the value it loaded back: no lane-dependent branch)
          if (r < TY && i < i1) {
            T* o = a.lamE1 + c3;
            __builtin_nontemporal_store(En[r][0], o);
            __builtin_nontemporal_store(En[r][1], o + 1);
            __builtin_nontemporal_store(En[r][2], o + 2);
          }
        }
      }
    }
    if (i > i0) {
      // new lamH of plane i-1, owned rows
      const int ih = i - 1;
#pragma unroll
      for (int r = 0; r < TY; ++r) {
        const int j = j0 + r;
        // new lamE of the cell after this one: the next lane's, or the next wave's first lane
        T ax = __shfl_down(Ep[r][0], 1), ay = __shfl_down(Ep[r][1], 1);
        if (lane == 63) { ax = xch[(i - 1) & 1][w + 1][r][0]; ay = xch[(i - 1) & 1][w + 1][r][1]; }
        // (cells of the hole included: adj_shell_E, which follows this launch, overwrites
        // them -- a test per row here cost 10 % of the launch)
        if (zin && j < by1) {
          const bool oy1 = j < ny - 1;
          // undo the E half step at (i-1, j, k): gather from the cell and its three high neighbours
          const T vx[3] = {zero, En[r][1], En[r][2]};
          const T vy[3] = {Ep[r + 1][0], zero, Ep[r + 1][2]};
          const T vz[3] = {ax, ay, zero};
          T g[6], gx[6], gy[6], gz[6];
          adj_march_G<T, true>(kE, Ep[r], ih == 0 && !a.open_lo, j == 0, k == 0, g);
          adj_march_G<T, true>(kE, vx, false, false, false, gx);
          adj_march_G<T, true>(kE, vy, false, false, false, gy);
          adj_march_G<T, true>(kE, vz, false, false, false, gz);
          T add[3];
          add[0] = (g[WY_X] - (oy1 ? gy[WY_X] : zero)) + (g[WZ_X] - (hasz1 ? gz[WZ_X] : zero));
          add[1] = (g[WX_Y] - (plane ? gx[WX_Y] : zero)) + (g[WZ_Y] - (hasz1 ? gz[WZ_Y] : zero));
          add[2] = (g[WX_Z] - (plane ? gx[WX_Z] : zero)) + (g[WY_Z] - (oy1 ? gy[WY_Z] : zero));
          const int64_t c3 = (int64_t)ih * sx + (int64_t)j * sy + 3 * k;
          T* o = a.lamH1 + c3;
          __builtin_nontemporal_store(Hp[r][0] + add[0], o);
          __builtin_nontemporal_store(Hp[r][1] + add[1], o + 1);
          __builtin_nontemporal_store(Hp[r][2] + add[2], o + 2);
        }
      }
    }
#pragma unroll
    for (int r = 0; r <= TY; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) { Hp[r][c] = Hc[r][c]; Ep[r][c] = En[r][c]; }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < TY; ++r) { xch[i & 1][w][r][0] = En[r][0]; xch[i & 1][w][r][1] = En[r][1]; }
    }
    __syncthreads();
  }
}

// The companions of adj_march_box for the cells next to PML slabs and objects: the adjoint of
// one half step over boxes of cells in ONE pass, out of place.  The two-pass form
// (adj_*_cells, adj_gather) keeps the six difference adjoints of every cell in `work`
// because pass 1 changes what it reads -- the memory-variable adjoints in place.  Here those
// go from one set of arrays to another (PmlK::psi -> psi_out), so a lane can evaluate pass 1
// of its own cell (with its side effects: memory-variable adjoints, object gradients) and,
// without side effects, of the three neighbours its gather reaches -- each only as far as the
// differences along that direction are concerned -- and sum them as adj_gather_cell does.
// Same operations on the same operands: equal to the two-pass form.  13 values per cell and
// half step through memory instead of 25.
//   H half step:  lamG = lamE:  lamE1(x) = lamE0(x) + sum over x, x - e of G_H
//   E half step:  lamG = lamH:  lamH1(x) = lamH0(x) + sum over x, x + e of G_E
// a.lamF: the adjoint of the field the half step advanced (read only; the rescaling of lamE
// under absorbing objects, the one change pass 1 makes to it, is left to adj_rescale_absorbing
// once every lane has read it).
template <typename T>
__global__ __launch_bounds__(256) void adj_shell_H(AdjArgs<T> a, const T* __restrict__ lamG_in,
                                                   T* __restrict__ lamG_out, CellBoxes box) {
  int i, j, k;
  if (!box_lane_cell(box, i, j, k)) return;
  const int64_t sy = 3 * (int64_t)a.nz, sx = sy * a.ny;
  const int64_t c3 = (int64_t)i * sx + (int64_t)j * sy + 3 * k;
  const bool ox = i > 0, oy = j > 0, oz = k > 0;
  const T* l = a.lamF + c3;
  const T* lx = ox ? l - sx : l;
  const T* ly = oy ? l - sy : l;
  const T* lz = oz ? l - 3 : l;
  const T own[3] = {l[0], l[1], l[2]};
  const T vx[3] = {lx[0], lx[1], lx[2]};
  const T vy[3] = {ly[0], ly[1], ly[2]};
  const T vz[3] = {lz[0], lz[1], lz[2]};
  const T o[3] = {lamG_in[c3], lamG_in[c3 + 1], lamG_in[c3 + 2]};
  T w[6], wx[6], wy[6], wz[6];
  adj_H_cell<T>(a, i, j, k, c3, own, w, true);
  adj_H_cell<T, 0>(a, ox ? i - 1 : i, j, k, ox ? c3 - sx : c3, vx, wx, false);
  adj_H_cell<T, 1>(a, i, oy ? j - 1 : j, k, oy ? c3 - sy : c3, vy, wy, false);
  adj_H_cell<T, 2>(a, i, j, oz ? k - 1 : k, oz ? c3 - 3 : c3, vz, wz, false);
  const T zero = T(0);
  // plane 0 of an x-slab: the two entries of the cell below come from the rank below
  const bool ghost = !ox && a.open_lo && a.wghost;
  const int64_t gr = ((int64_t)j * a.nz + k) * 2;
  const T bxy = ghost ? a.wghost[gr] : zero, bxz = ghost ? a.wghost[gr + 1] : zero;
  T add[3];
  add[0] = ((oy ? wy[WY_X] : zero) - w[WY_X]) + ((oz ? wz[WZ_X] : zero) - w[WZ_X]);
  add[1] = ((ox ? wx[WX_Y] : bxy) - w[WX_Y]) + ((oz ? wz[WZ_Y] : zero) - w[WZ_Y]);
  add[2] = ((ox ? wx[WX_Z] : bxz) - w[WX_Z]) + ((oy ? wy[WY_Z] : zero) - w[WY_Z]);
  lamG_out[c3] = o[0] + add[0];
  lamG_out[c3 + 1] = o[1] + add[1];
  lamG_out[c3 + 2] = o[2] + add[2];
}

template <typename T>
__global__ __launch_bounds__(256) void adj_shell_E(AdjArgs<T> a, const T* __restrict__ lamG_in,
                                                   T* __restrict__ lamG_out, CellBoxes box) {
  int i, j, k;
  if (!box_lane_cell(box, i, j, k)) return;
  const int64_t sy = 3 * (int64_t)a.nz, sx = sy * a.ny;
  const int64_t c3 = (int64_t)i * sx + (int64_t)j * sy + 3 * k;
  const bool ox = i < a.nx - 1, oy = j < a.ny - 1, oz = k < a.nz - 1;
  const T* l = a.lamF + c3;
  const T* lx = ox ? l + sx : l;
  const T* ly = oy ? l + sy : l;
  const T* lz = oz ? l + 3 : l;
  T own[3] = {l[0], l[1], l[2]};
  T vx[3] = {lx[0], lx[1], lx[2]};
  T vy[3] = {ly[0], ly[1], ly[2]};
  T vz[3] = {lz[0], lz[1], lz[2]};
  const T o[3] = {lamG_in[c3], lamG_in[c3 + 1], lamG_in[c3 + 2]};
  T w[6], wx[6], wy[6], wz[6];
  adj_E_cell<T>(a, i, j, k, c3, own, w, true);
  adj_E_cell<T, 0>(a, ox ? i + 1 : i, j, k, ox ? c3 + sx : c3, vx, wx, false);
  adj_E_cell<T, 1>(a, i, oy ? j + 1 : j, k, oy ? c3 + sy : c3, vy, wy, false);
  adj_E_cell<T, 2>(a, i, j, oz ? k + 1 : k, oz ? c3 + 3 : c3, vz, wz, false);
  const T zero = T(0);
  // the last plane of an x-slab: the two entries of the cell above come from the rank above
  const bool ghost = !ox && a.open_hi && a.wghost;
  const int64_t gr = ((int64_t)j * a.nz + k) * 2;
  const T bxy = ghost ? a.wghost[gr] : zero, bxz = ghost ? a.wghost[gr + 1] : zero;
  T add[3];
  add[0] = (w[WY_X] - (oy ? wy[WY_X] : zero)) + (w[WZ_X] - (oz ? wz[WZ_X] : zero));
  add[1] = (w[WX_Y] - (ox ? wx[WX_Y] : bxy)) + (w[WZ_Y] - (oz ? wz[WZ_Y] : zero));
  add[2] = (w[WX_Z] - (ox ? wx[WX_Z] : bxz)) + (w[WY_Z] - (oy ? wy[WY_Z] : zero));
  lamG_out[c3] = o[0] + add[0];
  lamG_out[c3 + 1] = o[1] + add[1];
  lamG_out[c3 + 2] = o[2] + add[2];
}

// Pass 2 over boxes of cells, out of place: adj_gather's sums with lamG_out = lamG_in + ...
// For the box around the objects, whose cells keep the two passes in the E half step (pass 1,
// adj_E_cells, is one evaluation per cell where adj_shell_E makes four, each a chain of
// dependent loads through the object table: 35 us against 12 + 8 us for a 66^3 box) and are
// all plain in the H half step (objects do not act there: no pass 1, the direct form).
template <typename T, bool PHASE_E>
__global__ __launch_bounds__(256) void adj_shell_gather(AdjArgs<T> a, const T* __restrict__ lamG_in,
                                                        T* __restrict__ lamG_out, CellBoxes box) {
  int i, j, k;
  if (!box_lane_cell(box, i, j, k)) return;
  const int64_t c3 = (((int64_t)i * a.ny + j) * a.nz + k) * 3;
  const T o[3] = {lamG_in[c3], lamG_in[c3 + 1], lamG_in[c3 + 2]};
  T add[3];
  if (a.direct && (PHASE_E ? adj_box_plain(a, i, i + 1, j, j + 1, k, k + 1)
                           : adj_box_plain(a, i - 1, i, j - 1, j, k - 1, k)))
    adj_gather_direct<T, PHASE_E>(a, i, j, k, add);
  else
    adj_gather_cell<T, PHASE_E>(a, i, j, k, add);
  lamG_out[c3] = o[0] + add[0];
  lamG_out[c3 + 1] = o[1] + add[1];
  lamG_out[c3 + 2] = o[2] + add[2];
}

// ---- the same for PML slabs without objects, every load requested up front ---------------
// adj_shell_H / adj_shell_E walk the plan's PML table cell by cell, four times per lane: chains
// of dependent round trips (descriptor, memory variables, profiles) that leave the memory
// system idle.  Most shell cells lie in slabs that no object comes near; for them the faces
// are kernel arguments (at most one low and one high face per axis, not overlapping), a lane
// finds the face of each axis by comparison, requests the field adjoints and every
// memory-variable adjoint it needs -- its own cell: one face per axis; a neighbour: the face
// normal to the direction it lies in -- and only then computes.  The arithmetic is that of
// adj_pml_face / adj_E_cell / adj_H_cell, operation for operation.  (512^3 float32, six
// ten-cell faces: 0.39 + 0.52 -> 0.27 + 0.26 ms per reverse step.  Loads without any test --
// a lane outside every face of an axis reading a dummy element -- cost two waves per SIMD in
// registers and were slower: 0.33 + 0.32 ms.)
template <typename T>
struct AdjFaceK {
  const T* psi;   // (2, slab) adjoints of the memory variables: read
  T* psi_out;     // written (the other set of arrays)
  const T* b;     // (t)
  const T* c;     // (t)
  int lo, t;      // index range [lo, lo + t) along the axis; t == 0: no such face
  int sy, sz;     // slab extents along y and z
};
template <typename T>
struct AdjFaceSet {
  AdjFaceK<T> f[3][2];   // [axis][0 low, 1 high]
};

template <typename T>
struct AdjFaceHit {
  bool inside, gate;
  int64_t s, slab;
  T* out;
  T p0, p1, cl, bl;
};

// The face normal to axis A that holds cell (i, j, k), if any, and its loads.  `valid`: the
// cell exists (a neighbour beyond the grid does not).
template <typename T, int A, bool PHASE_E, bool OWN>
__device__ __forceinline__ void adj_face_find(const AdjFaceSet<T>& fs, int nx, int i, int j, int k,
                                              bool valid, AdjFaceHit<T>& h) {
  const int q = A == 0 ? i : (A == 1 ? j : k);
  const bool in_lo = q >= fs.f[A][0].lo && q < fs.f[A][0].lo + fs.f[A][0].t;
  const bool in_hi = q >= fs.f[A][1].lo && q < fs.f[A][1].lo + fs.f[A][1].t;
  h.inside = valid && (in_lo || in_hi);
  const int lo = in_hi ? fs.f[A][1].lo : fs.f[A][0].lo, t = in_hi ? fs.f[A][1].t : fs.f[A][0].t;
  const int sy = in_hi ? fs.f[A][1].sy : fs.f[A][0].sy, sz = in_hi ? fs.f[A][1].sz : fs.f[A][0].sz;
  const T* psi = in_hi ? fs.f[A][1].psi : fs.f[A][0].psi;
  const T* cc = in_hi ? fs.f[A][1].c : fs.f[A][0].c;
  const int l = q - lo;
  const int li = A == 0 ? l : i, lj = A == 1 ? l : j, lk = A == 2 ? l : k;
  h.s = ((int64_t)li * sy + lj) * sz + lk;
  h.slab = (int64_t)(A == 0 ? t : nx) * sy * sz;
  h.gate = PHASE_E ? (l >= 1) : (l <= t - 2);
  h.p0 = h.p1 = h.cl = h.bl = T(0);
  h.out = nullptr;
  if (h.inside) {
    h.p0 = psi[h.s];
    h.p1 = psi[h.slab + h.s];
    h.cl = cc[l];
    if (OWN) {
      const T* bb = in_hi ? fs.f[A][1].b : fs.f[A][0].b;
      h.bl = bb[l];
      h.out = in_hi ? fs.f[A][1].psi_out : fs.f[A][0].psi_out;
    }
  }
}

// adj_pml_face on loaded values (a fused face): the two difference adjoints along A gain the
// memory-variable terms; OWN: the advanced memory-variable adjoints are stored.
template <typename T, int A, bool PHASE_E, bool OWN>
__device__ __forceinline__ void adj_face_apply(const AdjFaceHit<T>& h, T sc, const T lam[3], const T m[3],
                                               T& GdFu, T& GdFv) {
  constexpr int u = (A + 1) % 3, v = (A + 2) % 3;
  if (!h.inside) return;
  T lp0 = h.p0, lp1 = h.p1;
  const T sg = PHASE_E ? T(1) : T(-1);
  lp0 += sg * (-(sc * m[u])) * lam[u];
  lp1 += sg * (sc * m[v]) * lam[v];
  if (h.gate) {
    GdFv += h.cl * lp0;
    GdFu += h.cl * lp1;
  }
  if (OWN) {
    h.out[h.s] = h.bl * lp0;
    h.out[h.slab + h.s] = h.bl * lp1;
  }
}

template <typename T, bool PHASE_E>
__global__ __launch_bounds__(256) void adj_shell_faces(AdjArgs<T> a, AdjFaceSet<T> fs,
                                                       const T* __restrict__ lamG_in,
                                                       T* __restrict__ lamG_out, CellBoxes box) {
  int i, j, k;
  if (!box_lane_cell(box, i, j, k)) return;
  const int64_t sy = 3 * (int64_t)a.nz, sx = sy * a.ny;
  const int64_t c3 = (int64_t)i * sx + (int64_t)j * sy + 3 * k;
  constexpr int s = PHASE_E ? 1 : -1;
  const bool ox = PHASE_E ? i < a.nx - 1 : i > 0;
  const bool oy = PHASE_E ? j < a.ny - 1 : j > 0;
  const bool oz = PHASE_E ? k < a.nz - 1 : k > 0;
  const T* l = a.lamF + c3;
  const T* lx = ox ? l + s * sx : l;
  const T* ly = oy ? l + s * sy : l;
  const T* lz = oz ? l + s * 3 : l;
  // every load first
  const T own[3] = {l[0], l[1], l[2]};
  const T vx[3] = {lx[0], lx[1], lx[2]};
  const T vy[3] = {ly[0], ly[1], ly[2]};
  const T vz[3] = {lz[0], lz[1], lz[2]};
  const T o[3] = {lamG_in[c3], lamG_in[c3 + 1], lamG_in[c3 + 2]};
  AdjFaceHit<T> h0, h1, h2, nx_, ny_, nz_;
  adj_face_find<T, 0, PHASE_E, true>(fs, a.nx, i, j, k, true, h0);
  adj_face_find<T, 1, PHASE_E, true>(fs, a.nx, i, j, k, true, h1);
  adj_face_find<T, 2, PHASE_E, true>(fs, a.nx, i, j, k, true, h2);
  adj_face_find<T, 0, PHASE_E, false>(fs, a.nx, i + s, j, k, ox, nx_);
  adj_face_find<T, 1, PHASE_E, false>(fs, a.nx, i, j + s, k, oy, ny_);
  adj_face_find<T, 2, PHASE_E, false>(fs, a.nx, i, j, k + s, oz, nz_);
  const T m[3] = {a.bg[0], a.bg[1], a.bg[2]};
  const T zero = T(0);
  // own cell: adj_E_cell / adj_H_cell without objects
  T w[6];
#pragma unroll
  for (int n = 0; n < 6; ++n) w[n] = zero;
  adj_face_apply<T, 0, PHASE_E, true>(h0, a.sc, own, m, w[WX_Y], w[WX_Z]);
  adj_face_apply<T, 1, PHASE_E, true>(h1, a.sc, own, m, w[WY_Z], w[WY_X]);
  adj_face_apply<T, 2, PHASE_E, true>(h2, a.sc, own, m, w[WZ_X], w[WZ_Y]);
  T g[3], gx[3], gy[3], gz[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (PHASE_E) {
      g[c] = zero; g[c] += (a.sc * m[c]) * own[c];
      gx[c] = zero; gx[c] += (a.sc * m[c]) * vx[c];
      gy[c] = zero; gy[c] += (a.sc * m[c]) * vy[c];
      gz[c] = zero; gz[c] += (a.sc * m[c]) * vz[c];
    } else {
      g[c] = -((a.sc * m[c]) * own[c]);
      gx[c] = -((a.sc * m[c]) * vx[c]);
      gy[c] = -((a.sc * m[c]) * vy[c]);
      gz[c] = -((a.sc * m[c]) * vz[c]);
    }
  }
  w[WY_Z] += g[0]; w[WZ_Y] -= g[0];
  w[WZ_X] += g[1]; w[WX_Z] -= g[1];
  w[WX_Y] += g[2]; w[WY_X] -= g[2];
  if (PHASE_E) {
    if (i == 0 && !a.open_lo) w[WX_Y] = w[WX_Z] = 0;
    if (j == 0) w[WY_X] = w[WY_Z] = 0;
    if (k == 0) w[WZ_X] = w[WZ_Y] = 0;
  } else {
    if (i == a.nx - 1 && !a.open_hi) w[WX_Y] = w[WX_Z] = 0;
    if (j == a.ny - 1) w[WY_X] = w[WY_Z] = 0;
    if (k == a.nz - 1) w[WZ_X] = w[WZ_Y] = 0;
  }
  // the neighbours: the two entries along their direction (never on the edge that drops them)
  T wxy = zero, wxz = zero, wyz = zero, wyx = zero, wzx = zero, wzy = zero;
  adj_face_apply<T, 0, PHASE_E, false>(nx_, a.sc, vx, m, wxy, wxz);
  adj_face_apply<T, 1, PHASE_E, false>(ny_, a.sc, vy, m, wyz, wyx);
  adj_face_apply<T, 2, PHASE_E, false>(nz_, a.sc, vz, m, wzx, wzy);
  wxy += gx[2]; wxz -= gx[1];
  wyz += gy[0]; wyx -= gy[2];
  wzx += gz[1]; wzy -= gz[0];
  // across an open face of an x-slab the two entries come from the neighbour rank's plane
  if (!ox && (PHASE_E ? a.open_hi : a.open_lo) && a.wghost) {
    const int64_t gr = ((int64_t)j * a.nz + k) * 2;
    wxy = a.wghost[gr];
    wxz = a.wghost[gr + 1];
  }
  const bool hx = ox || ((PHASE_E ? a.open_hi : a.open_lo) && a.wghost);
  T add[3];
  if (PHASE_E) {
    add[0] = (w[WY_X] - (oy ? wyx : zero)) + (w[WZ_X] - (oz ? wzx : zero));
    add[1] = (w[WX_Y] - (hx ? wxy : zero)) + (w[WZ_Y] - (oz ? wzy : zero));
    add[2] = (w[WX_Z] - (hx ? wxz : zero)) + (w[WY_Z] - (oy ? wyz : zero));
  } else {
    add[0] = ((oy ? wyx : zero) - w[WY_X]) + ((oz ? wzx : zero) - w[WZ_X]);
    add[1] = ((hx ? wxy : zero) - w[WX_Y]) + ((oz ? wzy : zero) - w[WZ_Y]);
    add[2] = ((hx ? wxz : zero) - w[WX_Z]) + ((oy ? wyz : zero) - w[WY_Z]);
  }
  lamG_out[c3] = o[0] + add[0];
  lamG_out[c3 + 1] = o[1] + add[1];
  lamG_out[c3 + 2] = o[2] + add[2];
}

// What a rank sends across an open face of its x-slab in a whole reverse step: the two
// difference adjoints along x (WX_Y, WX_Z) of every cell of plane `i`, packed (ny, nz, 2) --
// what pass 1 would have left in work[plane i][..., 0:2], evaluated without its side effects.
// Undoing the H half step the last plane's go to the rank above (a.lamF = lamH as the step
// being undone left it), undoing the E half step plane 0's go to the rank below (a.lamF = the
// new lamE, before any rescaling under absorbing objects).
template <typename T, bool PHASE_E>
__global__ __launch_bounds__(256) void adj_pack_plane(AdjArgs<T> a, int i, T* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t plane = (int64_t)a.ny * a.nz;
  // (no early return: adj_E_cell / adj_H_cell vote across the wave)
  const bool live = r < plane;
  const int64_t rc = live ? r : plane - 1;
  const int j = (int)(rc / a.nz), k = (int)(rc % a.nz);
  const int64_t c3 = ((int64_t)i * plane + rc) * 3;
  T lam[3] = {a.lamF[c3], a.lamF[c3 + 1], a.lamF[c3 + 2]};
  T G[6];
  if (PHASE_E) adj_E_cell<T, 0>(a, i, j, k, c3, lam, G, false);
  else adj_H_cell<T, 0>(a, i, j, k, c3, lam, G, false);
  if (live) {
    out[r * 2] = G[WX_Y];
    out[r * 2 + 1] = G[WX_Z];
  }
}

// What pass 1 of the E half step does to lamE itself: under absorbing objects, last
// registered first, lamE *= (1 - f) / (1 + f) (the transpose of reference
// fdtd/objects.py:212-214).  One lane per cell of the box that holds every object.
template <typename T>
__global__ __launch_bounds__(256) void adj_rescale_absorbing(AdjArgs<T> a, T* lamE) {
  const int bx = a.hull[1] - a.hull[0], by = a.hull[3] - a.hull[2], bz = a.hull[5] - a.hull[4];
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (bx <= 0 || by <= 0 || bz <= 0 || t >= (int64_t)bx * by * bz) return;
  const int k = a.hull[4] + (int)(t % bz), j = a.hull[2] + (int)((t / bz) % by),
            i = a.hull[0] + (int)(t / ((int64_t)bz * by));
  const int64_t c3 = (((int64_t)i * a.ny + j) * a.nz + k) * 3;
  T lam[3] = {lamE[c3], lamE[c3 + 1], lamE[c3 + 2]};
  bool changed = false;
  for (int o = a.nobj - 1; o >= 0; --o) {
    const ObjK<T> O = load_obj<T>(a, o);
    if (O.kind != 1 || i < O.x0 || i >= O.x1 || j < O.y0 || j >= O.y1 || k < O.z0 || k >= O.z1) continue;
    const int64_t oc =
        ((int64_t)(i - O.x0) * (O.y1 - O.y0) + (j - O.y0)) * (O.z1 - O.z0) + (k - O.z0);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const T f = O.absorb[oc * 3 + c];
      lam[c] *= (T(1) - f) / (T(1) + f);
    }
    changed = true;
  }
  if (changed) {
    lamE[c3] = lam[0];
    lamE[c3 + 1] = lam[1];
    lamE[c3 + 2] = lam[2];
  }
}

}  // namespace fdtdhip
