// Whole Yee step in one pass for large 3-D grids: a workgroup marches along x.
//
// Two half-step passes move 18 values per cell and step through HBM even with constant
// materials (each pass reads both fields and writes one).  Here a step reads E and H once
// and writes the new E and H once: 12 values.  Out of place (E0, H0 -> E1, H1), like the
// one-lane-per-cell step_fused of yee_step.hpp, but without its thirteen neighbour loads
// per lane:
//
//   * a workgroup owns TY consecutive rows (y) of whole z rows and walks a chunk of x
//     planes; wave w of the workgroup holds the z segment [64 w, 64 w + 64), so every load
//     of a wave is a contiguous row segment, as in the half-step row kernels;
//   * a lane keeps, for each of its rows, H and the new E of the previous plane in
//     registers.  The new E of plane i needs H of plane i (own cell, the row above -- the
//     lane's own previous row or one extra row loaded above the tile -- and the cell
//     before, re-read through L1) and H of plane i-1 (registers).  The new H of plane i-1
//     needs the new E of plane i-1 at the cell (registers), at the row below (the lane's
//     next row: the row below the last owned one is evaluated redundantly, E only), at the
//     cell after (the next lane's register by a lane shift; for a wave's last lane the
//     first lane of the next wave, handed over through LDS with one workgroup barrier per
//     plane) and of plane i (just computed);
//   * chunks of the x range start one plane early for H of the plane before (read) and
//     evaluate the new E of one plane past their end redundantly.
//
// Redundant evaluations use the same operations on the same operands as the owner's, so
// the result equals two half steps bit for bit; only owners store fields, PML memory
// variables (psi_E into the second buffer: neighbours still read the old one) and curl
// traces.  Per cell and step: (TY+2)/TY rows of H and (TY+1)/TY rows of E read, one row of
// each written, times (LX+1)/LX for the reads.
//
// MI355X, 512^3 float32 vacuum, constants: 1.18 ms per step against 1.66 ms for two
// half-step launches (TY = 4: 102 registers, 4 waves per SIMD; other tile heights: see
// launch_march; chunks of 16 to 128 planes within 3 %, a single chunk 2.6 ms).  The rows and
// planes that neighbouring tiles and chunks read twice meet in the L2 (tiles in bands per
// XCD, plain loads): rocprofv3 counts 6.59 GB of traffic per step against 6.44 GB for every
// value moved once (float64: 13.6 against 12.9).
#pragma once

#include "plan.hpp"
#include "yee_generic.hpp"
#include "yee_step.hpp"

namespace fdtdhip {

constexpr int MARCH_MAX_WAVES = 16;   // z rows of up to 1024 cells

// Results leave with non-temporal stores (nothing reads them before the next step); every
// load is a plain one.  Non-temporal loads of E -- each row is needed once per tile -- were
// the first choice and cost 7 % in float32 and 17 % in float64 (512^3 vacuum: 1.258 -> 1.18
// and 3.03 -> 2.53 ms per step with plain loads): the row below a tile is the first row of
// the next tile, and a line loaded non-temporally is gone from the L2 by the time that tile
// asks for it.  (Non-temporal loads of the H rows no other tile reads: 1.28 ms, the z
// neighbour's plain load of the same line follows at once; plain stores: 1.235 ms.)
template <typename T>
__device__ __forceinline__ void march_store(T v, T* o) {
  __builtin_nontemporal_store(v, o);
}

template <typename T>
struct MarchArgs {
  StepArgs<T> s;
  int lx;          // planes per chunk
  int band;        // tiles per XCD (grid x = 8 * band), or 0: tile = blockIdx.x
  int box[6];      // step_march_box: x0, x1, y0, y1, z0, z1 of the cells to advance
  int hole[6];     // step_march_box: a box inside `box` whose E half step is not this kernel's
  int chunk0;      // step_march_box: first chunk of planes of this launch (a launch may take a
                   // part of the chunks: the others belong to another launch of the same step)
};

// Additive point sources on the new E of one cell (reference fdtd/grid.py:293-295).
template <typename T>
__device__ __forceinline__ void march_sources(const StepArgs<T>& a, int64_t c3, T e[3]) {
#pragma unroll
  for (int n = 0; n < MAX_STEP_SOURCES; ++n) {
    if (n >= a.nsrc) break;
    const int64_t off = a.src_elem[n] - c3;
    const T v = a.src_value[n];
    if (__any(off >= 0 && off < 3)) {
#pragma unroll
      for (int x = 0; x < 3; ++x) e[x] = off == x ? e[x] + v : e[x];
    }
  }
}

// CONST: both material arrays were proven constant (no branch on it per cell).
template <typename T, int TY, bool HAS_PML, bool HAS_OBJ, bool CONST, int MW = MARCH_MAX_WAVES>
__global__ __launch_bounds__(64 * MW) void step_march(MarchArgs<T> ma) {
  const StepArgs<T>& a = ma.s;
  const BulkArgs<T>& A = a.e;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = w * 64 + lane;
  const int nx = A.nx, ny = A.ny, nz = A.nz;
  const bool kin = k < nz;
  // Workgroups are dealt to the 8 XCDs round robin; tile t goes to XCD t % 8 unless
  // re-mapped.  With `band` tiles per XCD, XCD x takes the tiles [x band, (x+1) band): the
  // rows a tile reads above and below its own were then fetched into the same L2 by the
  // neighbouring tiles, which run at about the same planes.
  const int tile = ma.band > 0 ? (int)(blockIdx.x & 7) * ma.band + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile * TY >= ny) return;
  const int j0 = tile * TY;
  const int i0 = blockIdx.y * ma.lx, i1 = min(i0 + ma.lx, nx);
  const int64_t sy = 3 * (int64_t)nz, sx = sy * ny;
  const T* __restrict__ H0 = A.G;
  const T* __restrict__ E0 = A.F;
  // new E (components x, y) of each wave's first lane, for the last lane of the wave before
  __shared__ T xch[2][MARCH_MAX_WAVES + 1][TY][2];
  T Hp[TY + 1][3], Ep[TY + 1][3];
#pragma unroll
  for (int r = 0; r <= TY; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) { Hp[r][c] = T(0); Ep[r][c] = T(0); }
  if (i0 > 0 && kin) {
#pragma unroll
    for (int r = 0; r <= TY; ++r) {
      const int j = j0 + r;
      if (j < ny) {
        const T* p = H0 + (int64_t)(i0 - 1) * sx + (int64_t)j * sy + 3 * k;
        Hp[r][0] = p[0]; Hp[r][1] = p[1]; Hp[r][2] = p[2];
      }
    }
  }
  const bool hasz = k > 0, hasz1 = k < nz - 1;
  for (int i = i0; i <= i1; ++i) {
    T Hc[TY + 1][3], En[TY + 1][3];
#pragma unroll
    for (int r = 0; r <= TY; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) { Hc[r][c] = T(0); En[r][c] = T(0); }
    const bool plane = i < nx;
    if (plane && kin) {
      const int64_t pbase = (int64_t)i * sx + 3 * k;
      // every load of the plane is requested before the first one is used
      T zx[TY + 1], zy[TY + 1], ux = T(0), uz = T(0);
#pragma unroll
      for (int r = 0; r <= TY; ++r) {
        const int j = j0 + r;
        zx[r] = zy[r] = T(0);
        if (j < ny) {
          const int64_t c3 = pbase + (int64_t)j * sy;
          Hc[r][0] = H0[c3]; Hc[r][1] = H0[c3 + 1]; Hc[r][2] = H0[c3 + 2];
          En[r][0] = E0[c3]; En[r][1] = E0[c3 + 1]; En[r][2] = E0[c3 + 2];
          const T* z = H0 + (hasz ? c3 - 3 : c3);
          zx[r] = z[0]; zy[r] = z[1];
        }
      }
      if (j0 > 0) {
        const T* u = H0 + pbase + (int64_t)(j0 - 1) * sy;
        ux = u[0]; uz = u[2];
      }
      const bool hasx = i > 0;
#pragma unroll
      for (int r = 0; r <= TY; ++r) {
        const int j = j0 + r;
        if (j >= ny) continue;
        const int64_t c3 = pbase + (int64_t)j * sy;
        const bool hasy = j > 0;
        const T yx = r == 0 ? ux : Hc[r == 0 ? 0 : r - 1][0];
        const T yz = r == 0 ? uz : Hc[r == 0 ? 0 : r - 1][2];
        Diffs<T> d;
        d.x_y = hasx ? Hc[r][1] - Hp[r][1] : T(0);
        d.x_z = hasx ? Hc[r][2] - Hp[r][2] : T(0);
        d.y_x = hasy ? Hc[r][0] - yx : T(0);
        d.y_z = hasy ? Hc[r][2] - yz : T(0);
        d.z_x = hasz ? Hc[r][0] - zx[r] : T(0);
        d.z_y = hasz ? Hc[r][1] - zy[r] : T(0);
        const bool owner = r < TY && i < i1;
        T m[3];
        if (CONST && !HAS_OBJ) { m[0] = A.bg[0]; m[1] = A.bg[1]; m[2] = A.bg[2]; }
        else cell_material<T, false, HAS_OBJ>(A, c3, i, j, k, m);
        finish_E<T, HAS_PML, HAS_OBJ>(A, i, j, k, d, En[r], m, Hc[r], owner);
        if (a.nsrc > 0) march_sources<T>(a, c3, En[r]);
        if (owner) {
          T* o = a.E1 + c3;
          march_store(En[r][0], o);
          march_store(En[r][1], o + 1);
          march_store(En[r][2], o + 2);
        }
      }
    }
    if (i > i0) {
      // new H of plane i-1, owned rows
      const int ih = i - 1;
#pragma unroll
      for (int r = 0; r < TY; ++r) {
        const int j = j0 + r;
        // new E of the cell after this one: the next lane's, or the next wave's first lane
        T ax = __shfl_down(Ep[r][0], 1), ay = __shfl_down(Ep[r][1], 1);
        if (lane == 63) { ax = xch[(i - 1) & 1][w + 1][r][0]; ay = xch[(i - 1) & 1][w + 1][r][1]; }
        if (kin && j < ny) {
          const bool hasy1 = j < ny - 1;
          Diffs<T> d;
          d.x_y = plane ? En[r][1] - Ep[r][1] : T(0);
          d.x_z = plane ? En[r][2] - Ep[r][2] : T(0);
          d.y_x = hasy1 ? Ep[r + 1][0] - Ep[r][0] : T(0);
          d.y_z = hasy1 ? Ep[r + 1][2] - Ep[r][2] : T(0);
          d.z_x = hasz1 ? ax - Ep[r][0] : T(0);
          d.z_y = hasz1 ? ay - Ep[r][1] : T(0);
          const int64_t c3 = (int64_t)ih * sx + (int64_t)j * sy + 3 * k;
          T h[3] = {Hp[r][0], Hp[r][1], Hp[r][2]};
          T m[3];
          if (CONST) { m[0] = a.h.bg[0]; m[1] = a.h.bg[1]; m[2] = a.h.bg[2]; }
          else cell_material<T, false, false>(a.h, c3, ih, j, k, m);
          finish_H<T, HAS_PML>(a.h, ih, j, k, d, h, m);
          T* o = a.H1 + c3;
          march_store(h[0], o);
          march_store(h[1], o + 1);
          march_store(h[2], o + 2);
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

// The marching step over a box of cells (plans with PML faces: the box no face reaches; the
// shell around it takes half steps of its own, see yee_shell.hpp).  As step_march, except at
// the high faces of the box: where step_march evaluates the new E of the row below a tile, of
// the plane past a chunk and -- at the end of a z row -- of nothing, this kernel, when that
// row, plane or cell lies outside the box, READS the new E from E1, where the shell's E half
// step has put it before this launch.  The low neighbours are old H, read from H0 as ever.
// Objects: the kernel carries no object code (with it: 130 registers, one workgroup per CU,
// slower than two half-step launches).  The box that holds every object (`hole`) is part of
// the shell for the E half step -- its cells read their new E from E1 like cells outside the
// box -- and of this kernel for the H half step, in which objects do not act (reference
// fdtd/objects.py:128-134).
template <typename T, int TY, bool HAS_OBJ, bool CONST, int MW = MARCH_MAX_WAVES>
__global__ __launch_bounds__(64 * MW) void step_march_box(MarchArgs<T> ma) {
  const StepArgs<T>& a = ma.s;
  const BulkArgs<T>& A = a.e;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = w * 64 + lane;
  const int nx = A.nx, ny = A.ny, nz = A.nz;
  const int bx0 = ma.box[0], bx1 = ma.box[1], by0 = ma.box[2], by1 = ma.box[3], bz0 = ma.box[4],
            bz1 = ma.box[5];
  // lanes that advance a cell; the lane after the last of them fetches its cell's new E
  const bool zin = k >= bz0 && k < bz1;
  const bool zedge = k == bz1 && bz1 < nz;
  const bool zhole = HAS_OBJ && k >= ma.hole[4] && k < ma.hole[5];
  const int tile = ma.band > 0 ? (int)(blockIdx.x & 7) * ma.band + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int j0 = by0 + tile * TY;
  if (j0 >= by1) return;
  const int i0 = bx0 + ((int)blockIdx.y + ma.chunk0) * ma.lx, i1 = min(i0 + ma.lx, bx1);
  const int64_t sy = 3 * (int64_t)nz, sx = sy * ny;
  const T* __restrict__ H0 = A.G;
  const T* __restrict__ E0 = A.F;
  const T* E1r = a.E1;    // cells of the shell only: not written by this launch
  __shared__ T xch[2][MARCH_MAX_WAVES + 1][TY][2];
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
  for (int i = i0; i <= i1; ++i) {
    T Hc[TY + 1][3], En[TY + 1][3];
#pragma unroll
    for (int r = 0; r <= TY; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) { Hc[r][c] = T(0); En[r][c] = T(0); }
    const bool plane = i < nx;
    const bool pin = i < bx1;      // a plane of the box (else the plane past it, if any)
    bool holed[TY + 1];            // rows of this plane that cross the box of the objects
#pragma unroll
    for (int r = 0; r <= TY; ++r)
      holed[r] = HAS_OBJ && i >= ma.hole[0] && i < ma.hole[1] && j0 + r >= ma.hole[2] && j0 + r < ma.hole[3];
    if (plane && (zin || zedge)) {
      const int64_t pbase = (int64_t)i * sx + 3 * k;
      T zx[TY + 1], zy[TY + 1], ux = T(0), uz = T(0);
      // every load of the plane is requested before the first one is used
#pragma unroll
      for (int r = 0; r <= TY; ++r) {
        const int j = j0 + r;
        zx[r] = zy[r] = T(0);
        const int64_t c3 = pbase + (int64_t)j * sy;
        if (pin && zin && j < by1) {
          Hc[r][0] = H0[c3]; Hc[r][1] = H0[c3 + 1]; Hc[r][2] = H0[c3 + 2];
          if (zhole && holed[r]) {
            En[r][0] = E1r[c3]; En[r][1] = E1r[c3 + 1]; En[r][2] = E1r[c3 + 2];
          } else {
            En[r][0] = E0[c3]; En[r][1] = E0[c3 + 1]; En[r][2] = E0[c3 + 2];
            const T* z = H0 + (hasz ? c3 - 3 : c3);
            zx[r] = z[0]; zy[r] = z[1];
          }
        } else if (j < ny && ((j < by1 && r < TY) || (pin && zin && j == by1))) {
          // outside the box next to a cell of the tile: along x (the plane past the box),
          // along y (the row past it) or along z (the cell past it) -- the shell's new E
          En[r][0] = E1r[c3]; En[r][1] = E1r[c3 + 1]; En[r][2] = E1r[c3 + 2];
        }
      }
      if (pin && zin && j0 > 0) {
        const T* u = H0 + pbase + (int64_t)(j0 - 1) * sy;
        ux = u[0]; uz = u[2];
      }
      const bool hasx = i > 0;
      if (pin && zin) {
#pragma unroll
        for (int r = 0; r <= TY; ++r) {
          const int j = j0 + r;
          if (j >= by1 || (zhole && holed[r])) continue;
          const int64_t c3 = pbase + (int64_t)j * sy;
          const bool hasy = j > 0;
          const T yx = r == 0 ? ux : Hc[r == 0 ? 0 : r - 1][0];
          const T yz = r == 0 ? uz : Hc[r == 0 ? 0 : r - 1][2];
          Diffs<T> d;
          d.x_y = hasx ? Hc[r][1] - Hp[r][1] : T(0);
          d.x_z = hasx ? Hc[r][2] - Hp[r][2] : T(0);
          d.y_x = hasy ? Hc[r][0] - yx : T(0);
          d.y_z = hasy ? Hc[r][2] - yz : T(0);
          d.z_x = hasz ? Hc[r][0] - zx[r] : T(0);
          d.z_y = hasz ? Hc[r][1] - zy[r] : T(0);
          const bool owner = r < TY && i < i1;
          T m[3];
          // (no cell evaluated here lies under an object)
          if (CONST) { m[0] = A.bg[0]; m[1] = A.bg[1]; m[2] = A.bg[2]; }
          else cell_material<T, false, false>(A, c3, i, j, k, m);
          finish_E<T, false, false>(A, i, j, k, d, En[r], m, Hc[r], owner);
          if (a.nsrc > 0) march_sources<T>(a, c3, En[r]);
          if (owner) {
            T* o = a.E1 + c3;
            march_store(En[r][0], o);
            march_store(En[r][1], o + 1);
            march_store(En[r][2], o + 2);
          }
        }
      }
    }
    if (i > i0) {
      // new H of plane i-1, owned rows
      const int ih = i - 1;
#pragma unroll
      for (int r = 0; r < TY; ++r) {
        const int j = j0 + r;
        // new E of the cell after this one: the next lane's, or the next wave's first lane
        T ax = __shfl_down(Ep[r][0], 1), ay = __shfl_down(Ep[r][1], 1);
        if (lane == 63) { ax = xch[(i - 1) & 1][w + 1][r][0]; ay = xch[(i - 1) & 1][w + 1][r][1]; }
        if (zin && j < by1) {
          const bool hasy1 = j < ny - 1;
          Diffs<T> d;
          d.x_y = plane ? En[r][1] - Ep[r][1] : T(0);
          d.x_z = plane ? En[r][2] - Ep[r][2] : T(0);
          d.y_x = hasy1 ? Ep[r + 1][0] - Ep[r][0] : T(0);
          d.y_z = hasy1 ? Ep[r + 1][2] - Ep[r][2] : T(0);
          d.z_x = hasz1 ? ax - Ep[r][0] : T(0);
          d.z_y = hasz1 ? ay - Ep[r][1] : T(0);
          const int64_t c3 = (int64_t)ih * sx + (int64_t)j * sy + 3 * k;
          T h[3] = {Hp[r][0], Hp[r][1], Hp[r][2]};
          T m[3];
          if (CONST) { m[0] = a.h.bg[0]; m[1] = a.h.bg[1]; m[2] = a.h.bg[2]; }
          else cell_material<T, false, false>(a.h, c3, ih, j, k, m);
          finish_H<T, false>(a.h, ih, j, k, d, h, m);
          T* o = a.H1 + c3;
          march_store(h[0], o);
          march_store(h[1], o + 1);
          march_store(h[2], o + 2);
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

}  // namespace fdtdhip
