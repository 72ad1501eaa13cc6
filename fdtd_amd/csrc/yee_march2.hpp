// Two whole Yee steps in one pass: a wave marches along x and carries both steps in registers.
//
// step_march (yee_march.hpp) moves every field value through HBM once per step and spends its
// time waiting for them.  A box of cells with nothing between its steps but additive point
// sources (constant materials; PML faces, objects and slab faces at least three cells away and
// stepped by other launches -- which also record the curl traces of a differentiable run --,
// detector cells read by the patch kernel at the end of this file; no periodic boundary) can
// take two steps per pass: E and H are read once and written once per TWO steps, the state after the
// first one lives only in registers (reference fdtd/grid.py:267-295 run twice; the same sums on
// the same operands, so the result equals two single steps bit for bit).
//
//   * a wave owns TY rows (y) of a segment of 61 cells of a z row and walks a chunk of x
//     planes; its 64 lanes hold the cells [61 s - 1, 61 s + 63) of the row: the second step's
//     z neighbours are the first step's results of the neighbouring lanes, which leaves one
//     lane below and two above the owned ones without a valid result.  Waves share nothing and
//     never wait for each other (no LDS, no barrier); the lanes two neighbouring waves both hold
//     are evaluated twice from loads that meet in the L1;
//   * per plane i the wave loads H (TY+4 rows) and E (TY+3 rows) of the old state and then
//     evaluates, each from the previous stage's registers,
//         E' of plane i      rows j0-1 .. j0+TY+1     (E' = E after the first step)
//         H' of plane i-1    rows j0-1 .. j0+TY
//         E'' of plane i-1   rows j0   .. j0+TY       (stored: owned rows, owned lanes)
//         H'' of plane i-2   rows j0   .. j0+TY-1     (stored)
//     carrying H and E' of plane i, H' and E'' of plane i-1 to the next plane;
//   * a chunk of planes [i0, i1) starts at plane i0-1 with H of plane i0-2 pre-loaded and runs
//     to plane i1+1: three planes of loads and redundant evaluation per chunk.
//
// Every stage is evaluated for every row at every plane, valid or not: rows and planes outside
// the grid read clamped addresses, and nothing valid ever takes an operand from them (the
// one-sided difference rules at the faces of the grid, reference fdtd/grid.py:42-49, 67-74, cut
// the same links); only the stores are predicated.  The march comes in two versions, chosen per
// wave: one for the waves at a y face of the grid or with a source point among their cells, and
// a plain one without the tests of rows and sources, whose plane step is one basic block apart
// from its stores -- with those tests as branches the kernel spent four of five vector
// instructions on them and on spilled scalars.
//
// Per cell and two steps: (TY+4)/TY rows of H and (TY+3)/TY rows of E read (the rows of the
// neighbouring tiles meet in the L2: tiles are dealt to the XCDs in bands), one row of each
// written, times 64/61 along z and (LX+3)/LX along x.
#pragma once

#include "yee_march.hpp"

namespace fdtdhip {

constexpr int MARCH2_ZOWN = 61;    // cells of a z row a wave owns
constexpr int MARCH2_WAVES = 4;    // waves per workgroup (independent of each other)

template <typename T>
struct March2Args {
  const T* E0;                     // state before the two steps
  const T* H0;
  T* E1;                           // state after them
  T* H1;
  T ce[3], ch[3];                  // Courant number x inverse permittivity / permeability
  int nx, ny, nz;
  int lx;                          // planes per chunk
  int band;                        // workgroups per XCD (grid x = 8 * band), or 0: plain order
  int tiles, nseg;                 // tiles of TY rows, z segments per row
  int box[6];                      // x0, x1, y0, y1, z0, z1 of the cells to advance (stored); the
                                   // two cells around the box on every side are read and, with
                                   // the cells of the box, evaluated as plain cells: they must
                                   // be inside the grid's faces or plain themselves
  int hole[6];                     // a box inside `box` (objects and the two layers around them),
                                   // or zeros.  The march evaluates its cells as the plain cells
                                   // they are not and stores the results like any others: the
                                   // launches that take those cells' two steps one after the
                                   // other overwrite them afterwards, and nothing valid depends
                                   // on them (a per-lane test in the stores instead cost 4.5 %:
                                   // the waves that carry tests pace their workgroups).  Only the
                                   // detector kernel below consults it
  int nsrc;                        // additive source points: cell, component, value per step
  int src_i[MAX_STEP_SOURCES], src_j[MAX_STEP_SOURCES], src_k[MAX_STEP_SOURCES], src_c[MAX_STEP_SOURCES];
  T src_v[2][MAX_STEP_SOURCES];
};

// base + a byte offset held in 32 bits: the address form of a scalar base register plus a
// vector offset, which costs no vector arithmetic per access.
template <typename T>
__device__ __forceinline__ const T* at_bytes(const T* base, unsigned bytes) {
  return (const T*)((const char*)base + bytes);
}
template <typename T>
__device__ __forceinline__ T* at_bytes(T* base, unsigned bytes) {
  return (T*)((char*)base + bytes);
}

// Results leave with non-temporal stores, as in the one-step kernels (plain stores: no
// difference, 512^3 float32 1.71 against 1.68 ms per launch).
template <typename T>
__device__ __forceinline__ void march2_store(T v, T* o) { march_store(v, o); }

// The value of the lane above (k + 1) / below (k - 1); lanes 63 / 0 receive nothing useful.
template <typename T>
__device__ __forceinline__ T lane_above(T v) { return __shfl_down(v, 1); }
template <typename T>
__device__ __forceinline__ T lane_below(T v) { return __shfl_up(v, 1); }

// Additive point sources on the new E of plane i (reference fdtd/grid.py:293-295), in the
// order of their registration.
template <typename T, int R0, int R1, int NR>
__device__ __forceinline__ void march2_sources(const March2Args<T>& a, int step, int i, int j0, int k,
                                               T (&e)[NR][3]) {
#pragma unroll
  for (int n = 0; n < MAX_STEP_SOURCES; ++n) {
    if (n >= a.nsrc || a.src_i[n] != i) continue;
    const T v = a.src_v[step][n];
#pragma unroll
    for (int r = R0; r < R1; ++r) {
      const bool here = a.src_j[n] == j0 - 2 + r && a.src_k[n] == k;
#pragma unroll
      for (int c = 0; c < 3; ++c) e[r][c] = here && a.src_c[n] == c ? e[r][c] + v : e[r][c];
    }
  }
}

// What a wave knows about its place: row r of its register arrays is row j0 - 2 + r of the grid.
template <typename T, int NR>
struct March2Wave {
  int j0, k, i0, i1;
  int by0, by1;              // rows of the box
  bool kown, hasz, hasz1;
  unsigned koff, zoff;       // the lane's cell and its z neighbour within a row, in bytes (clamped)
  unsigned joff[NR];         // the rows within a plane, in elements (clamped)
  int64_t sx;                // elements per plane
  T ce[3], ch[3];
};

// One plane of the march: the four stages around plane i.  EDGE: with the rules of the y faces
// of the grid and the source points; without, every row the wave touches is known to lie inside
// the grid with neighbours on both sides, and no source point acts on any of its cells.
template <typename T, int TY, bool EDGE>
__device__ __forceinline__ void march2_plane(const March2Args<T>& a, const March2Wave<T, TY + 4>& w, int i,
                                             T (&Hp)[TY + 4][3], T (&E1p)[TY + 4][3],
                                             T (&H1p)[TY + 4][3], T (&E2p)[TY + 4][3]) {
  constexpr int NR = TY + 4;
  const int nx = a.nx, ny = a.ny;
  T Hc[NR][3], En[NR][3], H1c[NR][3], E2c[NR][3];
  const bool plane = i < nx;
  auto hasy = [&](int r) { return !EDGE || w.j0 - 2 + r > 0; };
  auto hasy1 = [&](int r) { return !EDGE || w.j0 - 2 + r < ny - 1; };
  auto inside = [&](int r) { return w.j0 - 2 + r < w.by1; };   // (owned rows start at the box's first)
  // ---- E' of plane i (a plane past the grid: clamped loads, nothing valid reads the result) ----
  {
    const int64_t pbase = (int64_t)min(i, nx - 1) * w.sx;
    const T* hp = a.H0 + pbase;
    const T* ep = a.E0 + pbase;
    T zx[NR], zy[NR];
    // every load of the plane is requested before the first one is used
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const T* h = at_bytes(hp + w.joff[r], w.koff);
      Hc[r][0] = h[0]; Hc[r][1] = h[1]; Hc[r][2] = h[2];
      if (r >= 1) {
        const T* e = at_bytes(ep + w.joff[r], w.koff);
        En[r][0] = e[0]; En[r][1] = e[1]; En[r][2] = e[2];
        const T* z = at_bytes(hp + w.joff[r], w.zoff);
        zx[r] = z[0]; zy[r] = z[1];
      } else {
        En[r][0] = En[r][1] = En[r][2] = T(0);
      }
    }
    const bool hasx = i > 0;
#pragma unroll
    for (int r = 1; r < NR; ++r) {
      const T x_y = hasx ? Hc[r][1] - Hp[r][1] : T(0);
      const T x_z = hasx ? Hc[r][2] - Hp[r][2] : T(0);
      const T y_x = hasy(r) ? Hc[r][0] - Hc[r - 1][0] : T(0);
      const T y_z = hasy(r) ? Hc[r][2] - Hc[r - 1][2] : T(0);
      const T z_x = w.hasz ? Hc[r][0] - zx[r] : T(0);
      const T z_y = w.hasz ? Hc[r][1] - zy[r] : T(0);
      En[r][0] += w.ce[0] * (y_z - z_y);
      En[r][1] += w.ce[1] * (z_x - x_z);
      En[r][2] += w.ce[2] * (x_y - y_x);
    }
    if (EDGE && a.nsrc > 0) march2_sources<T, 1, NR, NR>(a, 0, i, w.j0, w.k, En);
  }
  // ---- H' of plane i-1 ------------------------------------------------------------------
  const int ih = i - 1;
#pragma unroll
  for (int r = 0; r < NR; ++r) { H1c[r][0] = H1c[r][1] = H1c[r][2] = T(0); }
#pragma unroll
  for (int r = 1; r < TY + 3; ++r) {
    const T ax = lane_above(E1p[r][0]), ay = lane_above(E1p[r][1]);
    const T x_y = plane ? En[r][1] - E1p[r][1] : T(0);
    const T x_z = plane ? En[r][2] - E1p[r][2] : T(0);
    const T y_x = hasy1(r) ? E1p[r + 1][0] - E1p[r][0] : T(0);
    const T y_z = hasy1(r) ? E1p[r + 1][2] - E1p[r][2] : T(0);
    const T z_x = w.hasz1 ? ax - E1p[r][0] : T(0);
    const T z_y = w.hasz1 ? ay - E1p[r][1] : T(0);
    H1c[r][0] = Hp[r][0] - w.ch[0] * (y_z - z_y);
    H1c[r][1] = Hp[r][1] - w.ch[1] * (z_x - x_z);
    H1c[r][2] = Hp[r][2] - w.ch[2] * (x_y - y_x);
  }
  // ---- E'' of plane i-1 -----------------------------------------------------------------
  {
    const bool hasx = ih > 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) { E2c[r][0] = E2c[r][1] = E2c[r][2] = T(0); }
#pragma unroll
    for (int r = 2; r < TY + 3; ++r) {
      const T zx = lane_below(H1c[r][0]), zy = lane_below(H1c[r][1]);
      const T x_y = hasx ? H1c[r][1] - H1p[r][1] : T(0);
      const T x_z = hasx ? H1c[r][2] - H1p[r][2] : T(0);
      const T y_x = hasy(r) ? H1c[r][0] - H1c[r - 1][0] : T(0);
      const T y_z = hasy(r) ? H1c[r][2] - H1c[r - 1][2] : T(0);
      const T z_x = w.hasz ? H1c[r][0] - zx : T(0);
      const T z_y = w.hasz ? H1c[r][1] - zy : T(0);
      E2c[r][0] = E1p[r][0] + w.ce[0] * (y_z - z_y);
      E2c[r][1] = E1p[r][1] + w.ce[1] * (z_x - x_z);
      E2c[r][2] = E1p[r][2] + w.ce[2] * (x_y - y_x);
    }
    if (EDGE && a.nsrc > 0) march2_sources<T, 2, TY + 3, NR>(a, 1, ih, w.j0, w.k, E2c);
    if (ih >= w.i0 && ih < w.i1 && w.kown) {
      T* ep = a.E1 + (int64_t)ih * w.sx;
#pragma unroll
      for (int r = 2; r < TY + 2; ++r) {
        if (!inside(r)) continue;
        T* o = at_bytes(ep + w.joff[r], w.koff);
        march2_store(E2c[r][0], o);
        march2_store(E2c[r][1], o + 1);
        march2_store(E2c[r][2], o + 2);
      }
    }
  }
  // ---- H'' of plane i-2 -----------------------------------------------------------------
  {
    const int ig = i - 2;
    const bool plane2 = ih < nx;     // E'' of plane i-1 exists
    T h[NR][3];
#pragma unroll
    for (int r = 2; r < TY + 2; ++r) {
      const T ax = lane_above(E2p[r][0]), ay = lane_above(E2p[r][1]);
      const T x_y = plane2 ? E2c[r][1] - E2p[r][1] : T(0);
      const T x_z = plane2 ? E2c[r][2] - E2p[r][2] : T(0);
      const T y_x = hasy1(r) ? E2p[r + 1][0] - E2p[r][0] : T(0);
      const T y_z = hasy1(r) ? E2p[r + 1][2] - E2p[r][2] : T(0);
      const T z_x = w.hasz1 ? ax - E2p[r][0] : T(0);
      const T z_y = w.hasz1 ? ay - E2p[r][1] : T(0);
      h[r][0] = H1p[r][0] - w.ch[0] * (y_z - z_y);
      h[r][1] = H1p[r][1] - w.ch[1] * (z_x - x_z);
      h[r][2] = H1p[r][2] - w.ch[2] * (x_y - y_x);
    }
    if (ig >= w.i0 && ig < w.i1 && w.kown) {
      T* hp = a.H1 + (int64_t)ig * w.sx;
#pragma unroll
      for (int r = 2; r < TY + 2; ++r) {
        if (!inside(r)) continue;
        T* o = at_bytes(hp + w.joff[r], w.koff);
        march2_store(h[r][0], o);
        march2_store(h[r][1], o + 1);
        march2_store(h[r][2], o + 2);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Hp[r][c] = Hc[r][c]; E1p[r][c] = En[r][c]; H1p[r][c] = H1c[r][c]; E2p[r][c] = E2c[r][c];
    }
}

template <typename T, int TY>
__global__ __launch_bounds__(64 * MARCH2_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void step_march2(March2Args<T> a) {
  constexpr int NR = TY + 4;
  const int lane = threadIdx.x & 63;
  const int group = a.band > 0 ? (int)(blockIdx.x & 7) * a.band + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int gw = group * MARCH2_WAVES + (int)(threadIdx.x >> 6);
  const int tile = gw / a.nseg, zseg = gw - tile * a.nseg;
  if (tile >= a.tiles) return;       // (whole waves leave: nothing in this kernel waits for them)
  const int nx = a.nx, ny = a.ny, nz = a.nz;
  March2Wave<T, NR> w;
  w.k = a.box[4] + zseg * MARCH2_ZOWN - 1 + lane;
  w.kown = w.k < a.box[5] && lane >= 1 && lane <= MARCH2_ZOWN;
  w.hasz = w.k > 0;
  w.hasz1 = w.k < nz - 1;
  w.by0 = a.box[2];
  w.by1 = a.box[3];
  w.j0 = w.by0 + tile * TY;
  w.i0 = a.box[0] + blockIdx.y * a.lx;
  w.i1 = min(w.i0 + a.lx, a.box[1]);
  const int is = max(w.i0 - 1, 0);   // first plane evaluated
  const unsigned sy = 3u * (unsigned)nz;
  w.sx = (int64_t)sy * ny;
  w.koff = 3u * (unsigned)sizeof(T) * (unsigned)min(max(w.k, 0), nz - 1);
  w.zoff = w.hasz ? w.koff - 3u * (unsigned)sizeof(T) : w.koff;
#pragma unroll
  for (int r = 0; r < NR; ++r) w.joff[r] = (unsigned)min(max(w.j0 - 2 + r, 0), ny - 1) * sy;
#pragma unroll
  for (int c = 0; c < 3; ++c) { w.ce[c] = a.ce[c]; w.ch[c] = a.ch[c]; }
  // a wave takes the version with the tests unless each of its rows and the two above and below
  // lie inside the grid and no source point is among the cells it evaluates
  bool edge = w.j0 < 3 || w.j0 + TY + 2 >= ny;
  // (the hole: rows and planes are tested per store; any lane in its z range will do here)
#pragma unroll
  for (int n = 0; n < MAX_STEP_SOURCES; ++n)
    edge |= n < a.nsrc && a.src_j[n] >= w.j0 - 2 && a.src_j[n] < w.j0 - 2 + NR &&
            a.src_k[n] >= a.box[4] + zseg * MARCH2_ZOWN - 1 && a.src_k[n] < a.box[4] + zseg * MARCH2_ZOWN + 63;
  T Hp[NR][3], E1p[NR][3], H1p[NR][3], E2p[NR][3];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) { Hp[r][c] = T(0); E1p[r][c] = T(0); H1p[r][c] = T(0); E2p[r][c] = T(0); }
  if (is > 0) {
    const T* hp = a.H0 + (int64_t)(is - 1) * w.sx;
#pragma unroll
    for (int r = 1; r < NR; ++r) {
      const T* p = at_bytes(hp + w.joff[r], w.koff);
      Hp[r][0] = p[0]; Hp[r][1] = p[1]; Hp[r][2] = p[2];
    }
  }
  const int iend = min(w.i1 + 1, nx + 1);
  if (edge) {
    for (int i = is; i <= iend; ++i) march2_plane<T, TY, true>(a, w, i, Hp, E1p, H1p, E2p);
  } else {
    for (int i = is; i <= iend; ++i) march2_plane<T, TY, false>(a, w, i, Hp, E1p, H1p, E2p);
  }
}


// ---- z rows of up to 512 cells: a workgroup per tile, every lane owns its cell -------------
//
// The independent waves above pay for their independence in memory requests: their windows of 61
// cells start at no cache line, so a row segment of 732 B touches seven 128-byte lines instead of
// six, the segments two waves store share a line, and a row of 512 cells takes nine waves (512^3
// float32: 1.68 ms per launch; with the stores left out 1.07 ms, with windows of 64 owned cells --
// wrong results, the cost of the addresses alone -- 1.30 ms).  Where all waves of a z row fit one
// workgroup (eight waves, each lane with up to 256 registers) they hand each other the three
// values per row and step that cross a wave's ends through LDS instead, at one barrier per plane:
// the stages are one more plane apart, so that what a stage needs from the neighbouring wave was
// written before the previous barrier,
//       E' of plane i,  H' of plane i-1,  E'' of plane i-2,  H'' of plane i-3,
// at the price of one more plane of E' and of H' carried in registers and a fourth plane of
// redundant evaluation per chunk.

constexpr int MARCH2_ROW_WAVES = 8;

__device__ __forceinline__ float wave_above(float v) {      // lane l receives lane l + 1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_below(float v) {      // lane l receives lane l - 1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ double wave_above(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_below(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// What crosses the ends of the waves, per parity of the plane: components x and y of each row.
// Slot v holds what wave v - 1's last lane and wave v's first lane wrote.
template <typename T, int TY, int MW>
struct March2Exchange {
  T e1[2][MW + 1][TY + 4][2];   // E' of the first lanes (for the last lane of the wave before)
  T h1[2][MW + 1][TY + 4][2];   // H' of the last lanes (for the first lane of the wave after)
  T e2[2][MW + 1][TY + 4][2];   // E'' of the first lanes
};

template <typename T, int TY, int MW, bool EDGE>
__device__ __forceinline__ void march2_row_plane(const March2Args<T>& a, const March2Wave<T, TY + 4>& w,
                                                 March2Exchange<T, TY, MW>& x, int wave, int lane, int i,
                                                 T (&Hp)[TY + 4][3], T (&E1a)[TY + 4][3], T (&E1b)[TY + 4][3],
                                                 T (&H1a)[TY + 4][3], T (&H1b)[TY + 4][3], T (&E2a)[TY + 4][3]) {
  constexpr int NR = TY + 4;
  const int nx = a.nx, ny = a.ny;
  T Hc[NR][3], En[NR][3], H1c[NR][3], E2c[NR][3];
  const bool plane = i < nx;
  const bool first = lane == 0, last = lane == 63;
  auto hasy = [&](int r) { return !EDGE || w.j0 - 2 + r > 0; };
  auto hasy1 = [&](int r) { return !EDGE || w.j0 - 2 + r < ny - 1; };
  auto inside = [&](int r) { return w.j0 - 2 + r < w.by1; };   // (owned rows start at the box's first)
  // ---- E' of plane i ----------------------------------------------------------------------
  // (requesting the loads of plane i+1 here, to travel while the other stages are evaluated:
  // 512^3 float32 1.74 against 1.57 ms per launch -- the registers they occupy, not the wait
  // for them, is what the kernel is short of)
  {
    const int64_t pbase = (int64_t)min(i, nx - 1) * w.sx;
    const T* hp = a.H0 + pbase;
    const T* ep = a.E0 + pbase;
    T zx[NR], zy[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const T* h = at_bytes(hp + w.joff[r], w.koff);
      Hc[r][0] = h[0]; Hc[r][1] = h[1]; Hc[r][2] = h[2];
      if (r >= 1) {
        const T* e = at_bytes(ep + w.joff[r], w.koff);
        En[r][0] = e[0]; En[r][1] = e[1]; En[r][2] = e[2];
        const T* z = at_bytes(hp + w.joff[r], w.zoff);
        zx[r] = z[0]; zy[r] = z[1];
      } else {
        En[r][0] = En[r][1] = En[r][2] = T(0);
      }
    }
    const bool hasx = i > 0;
#pragma unroll
    for (int r = 1; r < NR; ++r) {
      const T x_y = hasx ? Hc[r][1] - Hp[r][1] : T(0);
      const T x_z = hasx ? Hc[r][2] - Hp[r][2] : T(0);
      const T y_x = hasy(r) ? Hc[r][0] - Hc[r - 1][0] : T(0);
      const T y_z = hasy(r) ? Hc[r][2] - Hc[r - 1][2] : T(0);
      const T z_x = w.hasz ? Hc[r][0] - zx[r] : T(0);
      const T z_y = w.hasz ? Hc[r][1] - zy[r] : T(0);
      En[r][0] += w.ce[0] * (y_z - z_y);
      En[r][1] += w.ce[1] * (z_x - x_z);
      En[r][2] += w.ce[2] * (x_y - y_x);
    }
    if (EDGE && a.nsrc > 0) march2_sources<T, 1, NR, NR>(a, 0, i, w.j0, w.k, En);
    if (first) {
#pragma unroll
      for (int r = 1; r < TY + 3; ++r) { x.e1[i & 1][wave][r][0] = En[r][0]; x.e1[i & 1][wave][r][1] = En[r][1]; }
    }
  }
  // ---- H' of plane i-1 ----------------------------------------------------------------------
#pragma unroll
  for (int r = 0; r < NR; ++r) { H1c[r][0] = H1c[r][1] = H1c[r][2] = T(0); }
#pragma unroll
  for (int r = 1; r < TY + 3; ++r) {
    const T nx0 = x.e1[(i - 1) & 1][wave + 1][r][0], ny0 = x.e1[(i - 1) & 1][wave + 1][r][1];
    const T sx0 = wave_above(E1a[r][0]), sy0 = wave_above(E1a[r][1]);
    const T ax = last ? nx0 : sx0, ay = last ? ny0 : sy0;
    const T x_y = plane ? En[r][1] - E1a[r][1] : T(0);
    const T x_z = plane ? En[r][2] - E1a[r][2] : T(0);
    const T y_x = hasy1(r) ? E1a[r + 1][0] - E1a[r][0] : T(0);
    const T y_z = hasy1(r) ? E1a[r + 1][2] - E1a[r][2] : T(0);
    const T z_x = w.hasz1 ? ax - E1a[r][0] : T(0);
    const T z_y = w.hasz1 ? ay - E1a[r][1] : T(0);
    H1c[r][0] = Hp[r][0] - w.ch[0] * (y_z - z_y);
    H1c[r][1] = Hp[r][1] - w.ch[1] * (z_x - x_z);
    H1c[r][2] = Hp[r][2] - w.ch[2] * (x_y - y_x);
  }
  if (last) {
#pragma unroll
    for (int r = 2; r < TY + 3; ++r) { x.h1[(i - 1) & 1][wave + 1][r][0] = H1c[r][0]; x.h1[(i - 1) & 1][wave + 1][r][1] = H1c[r][1]; }
  }
  // ---- E'' of plane i-2 ---------------------------------------------------------------------
  {
    const int ie = i - 2;
    const bool hasx = ie > 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) { E2c[r][0] = E2c[r][1] = E2c[r][2] = T(0); }
#pragma unroll
    for (int r = 2; r < TY + 3; ++r) {
      const T nx0 = x.h1[ie & 1][wave][r][0], ny0 = x.h1[ie & 1][wave][r][1];
      const T sx0 = wave_below(H1a[r][0]), sy0 = wave_below(H1a[r][1]);
      const T zx = first ? nx0 : sx0, zy = first ? ny0 : sy0;
      const T x_y = hasx ? H1a[r][1] - H1b[r][1] : T(0);
      const T x_z = hasx ? H1a[r][2] - H1b[r][2] : T(0);
      const T y_x = hasy(r) ? H1a[r][0] - H1a[r - 1][0] : T(0);
      const T y_z = hasy(r) ? H1a[r][2] - H1a[r - 1][2] : T(0);
      const T z_x = w.hasz ? H1a[r][0] - zx : T(0);
      const T z_y = w.hasz ? H1a[r][1] - zy : T(0);
      E2c[r][0] = E1b[r][0] + w.ce[0] * (y_z - z_y);
      E2c[r][1] = E1b[r][1] + w.ce[1] * (z_x - x_z);
      E2c[r][2] = E1b[r][2] + w.ce[2] * (x_y - y_x);
    }
    if (EDGE && a.nsrc > 0) march2_sources<T, 2, TY + 3, NR>(a, 1, ie, w.j0, w.k, E2c);
    if (first) {
#pragma unroll
      for (int r = 2; r < TY + 2; ++r) { x.e2[ie & 1][wave][r][0] = E2c[r][0]; x.e2[ie & 1][wave][r][1] = E2c[r][1]; }
    }
    if (ie >= w.i0 && ie < w.i1 && w.kown) {
      T* ep = a.E1 + (int64_t)ie * w.sx;
#pragma unroll
      for (int r = 2; r < TY + 2; ++r) {
        if (!inside(r)) continue;
        T* o = at_bytes(ep + w.joff[r], w.koff);
        march2_store(E2c[r][0], o);
        march2_store(E2c[r][1], o + 1);
        march2_store(E2c[r][2], o + 2);
      }
    }
  }
  // ---- H'' of plane i-3 ---------------------------------------------------------------------
  {
    const int ig = i - 3;
    const bool plane2 = i - 2 < nx;     // E'' of plane i-2 exists
    T h[NR][3];
#pragma unroll
    for (int r = 2; r < TY + 2; ++r) {
      const T nx0 = x.e2[ig & 1][wave + 1][r][0], ny0 = x.e2[ig & 1][wave + 1][r][1];
      const T sx0 = wave_above(E2a[r][0]), sy0 = wave_above(E2a[r][1]);
      const T ax = last ? nx0 : sx0, ay = last ? ny0 : sy0;
      const T x_y = plane2 ? E2c[r][1] - E2a[r][1] : T(0);
      const T x_z = plane2 ? E2c[r][2] - E2a[r][2] : T(0);
      const T y_x = hasy1(r) ? E2a[r + 1][0] - E2a[r][0] : T(0);
      const T y_z = hasy1(r) ? E2a[r + 1][2] - E2a[r][2] : T(0);
      const T z_x = w.hasz1 ? ax - E2a[r][0] : T(0);
      const T z_y = w.hasz1 ? ay - E2a[r][1] : T(0);
      h[r][0] = H1b[r][0] - w.ch[0] * (y_z - z_y);
      h[r][1] = H1b[r][1] - w.ch[1] * (z_x - x_z);
      h[r][2] = H1b[r][2] - w.ch[2] * (x_y - y_x);
    }
    if (ig >= w.i0 && ig < w.i1 && w.kown) {
      T* hp = a.H1 + (int64_t)ig * w.sx;
#pragma unroll
      for (int r = 2; r < TY + 2; ++r) {
        if (!inside(r)) continue;
        T* o = at_bytes(hp + w.joff[r], w.koff);
        march2_store(h[r][0], o);
        march2_store(h[r][1], o + 1);
        march2_store(h[r][2], o + 2);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Hp[r][c] = Hc[r][c];
      E1b[r][c] = E1a[r][c]; E1a[r][c] = En[r][c];
      H1b[r][c] = H1a[r][c]; H1a[r][c] = H1c[r][c];
      E2a[r][c] = E2c[r][c];
    }
  __syncthreads();
}

// Workgroup = the waves of the z rows of one tile (blockDim.x = 64 x waves <= 64 MW), grid x =
// tiles (in bands per XCD), grid y = chunks of planes.  MW = 8: up to 256 registers per lane;
// MW = 16 (z rows of up to 1024 cells): 128, which tiles of two rows fit in float32.
template <typename T, int TY, int MW = MARCH2_ROW_WAVES>
__global__ __launch_bounds__(64 * MW) void step_march2_rows(March2Args<T> a) {
  constexpr int NR = TY + 4;
  __shared__ March2Exchange<T, TY, MW> x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = a.band > 0 ? (int)(blockIdx.x & 7) * a.band + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile >= a.tiles) return;       // (the whole workgroup)
  const int nx = a.nx, ny = a.ny, nz = a.nz;
  March2Wave<T, NR> w;
  w.k = wave * 64 + lane;
  w.kown = w.k >= a.box[4] && w.k < a.box[5];
  w.hasz = w.k > 0;
  w.hasz1 = w.k < nz - 1;
  w.by0 = a.box[2];
  w.by1 = a.box[3];
  w.j0 = w.by0 + tile * TY;
  w.i0 = a.box[0] + blockIdx.y * a.lx;
  w.i1 = min(w.i0 + a.lx, a.box[1]);
  const int is = max(w.i0 - 1, 0);
  const unsigned sy = 3u * (unsigned)nz;
  w.sx = (int64_t)sy * ny;
  w.koff = 3u * (unsigned)sizeof(T) * (unsigned)min(w.k, nz - 1);
  w.zoff = w.hasz ? w.koff - 3u * (unsigned)sizeof(T) : w.koff;
#pragma unroll
  for (int r = 0; r < NR; ++r) w.joff[r] = (unsigned)min(max(w.j0 - 2 + r, 0), ny - 1) * sy;
#pragma unroll
  for (int c = 0; c < 3; ++c) { w.ce[c] = a.ce[c]; w.ch[c] = a.ch[c]; }
  bool edge = w.j0 < 3 || w.j0 + TY + 2 >= ny;
  // (the hole: rows and planes are tested per store; any lane in its z range will do here)
#pragma unroll
  for (int n = 0; n < MAX_STEP_SOURCES; ++n)
    edge |= n < a.nsrc && a.src_j[n] >= w.j0 - 2 && a.src_j[n] < w.j0 - 2 + NR &&
            a.src_k[n] >= wave * 64 && a.src_k[n] < wave * 64 + 64;
  T Hp[NR][3], E1a[NR][3], E1b[NR][3], H1a[NR][3], H1b[NR][3], E2a[NR][3];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Hp[r][c] = T(0); E1a[r][c] = T(0); E1b[r][c] = T(0); H1a[r][c] = T(0); H1b[r][c] = T(0); E2a[r][c] = T(0);
    }
  if (is > 0) {
    const T* hp = a.H0 + (int64_t)(is - 1) * w.sx;
#pragma unroll
    for (int r = 1; r < NR; ++r) {
      const T* p = at_bytes(hp + w.joff[r], w.koff);
      Hp[r][0] = p[0]; Hp[r][1] = p[1]; Hp[r][2] = p[2];
    }
  }
  // (every wave of the workgroup takes the same number of planes, whichever version it runs:
  // the barriers pair up)
  const int iend = min(w.i1 + 2, nx + 2);
  if (edge) {
    for (int i = is; i <= iend; ++i) march2_row_plane<T, TY, MW, true>(a, w, x, wave, lane, i, Hp, E1a, E1b, H1a, H1b, E2a);
  } else {
    for (int i = is; i <= iend; ++i) march2_row_plane<T, TY, MW, false>(a, w, x, wave, lane, i, Hp, E1a, E1b, H1a, H1b, E2a);
  }
}


// ---- detectors ----------------------------------------------------------------------------
// A detector reads E after the sources and H after the H half step of EVERY step (reference
// fdtd/detectors.py:114-124).  For the cells the two-step kernels store, the state after the
// first step of a pair exists in registers only; this kernel evaluates it once more for the few
// cells a detector reads -- plain cells with plain neighbours, from the state before the pair:
// the cell's E', and for its H' the E' of the three cells after it as well.  Same operations on
// the same operands as the owner's.  One lane per detector cell; the cells outside `box` or inside
// `hole` are left to the ordinary detector kernel, which reads them after the first step.

// E after the first step (sources included) of the plain cell (i, j, k).
template <typename T>
__device__ __forceinline__ void march2_plain_E1(const March2Args<T>& a, int i, int j, int k, T e[3]) {
  const int64_t sy = 3 * (int64_t)a.nz, sx = sy * a.ny;
  const int64_t c3 = (int64_t)i * sx + (int64_t)j * sy + 3 * k;
  const T* h = a.H0 + c3;
  const T* hx = i > 0 ? h - sx : h;
  const T* hy = j > 0 ? h - sy : h;
  const T* hz = k > 0 ? h - 3 : h;
  const T x_y = i > 0 ? h[1] - hx[1] : T(0), x_z = i > 0 ? h[2] - hx[2] : T(0);
  const T y_x = j > 0 ? h[0] - hy[0] : T(0), y_z = j > 0 ? h[2] - hy[2] : T(0);
  const T z_x = k > 0 ? h[0] - hz[0] : T(0), z_y = k > 0 ? h[1] - hz[1] : T(0);
  e[0] = a.E0[c3] + a.ce[0] * (y_z - z_y);
  e[1] = a.E0[c3 + 1] + a.ce[1] * (z_x - x_z);
  e[2] = a.E0[c3 + 2] + a.ce[2] * (x_y - y_x);
#pragma unroll
  for (int n = 0; n < MAX_STEP_SOURCES; ++n)
    if (n < a.nsrc && a.src_i[n] == i && a.src_j[n] == j && a.src_k[n] == k) {
#pragma unroll
      for (int c = 0; c < 3; ++c) e[c] = a.src_c[n] == c ? e[c] + a.src_v[0][n] : e[c];
    }
}

constexpr int MARCH2_MAX_DETS = 8;
template <typename T>
struct March2Dets {
  int count;
  const int64_t* cell[MARCH2_MAX_DETS];   // flat cell indices of each detector
  T* outE[MARCH2_MAX_DETS];               // its record rows of the first step
  T* outH[MARCH2_MAX_DETS];
  int64_t first[MARCH2_MAX_DETS + 1];     // in cells
};

template <typename T>
__global__ void march2_detect_first_step(March2Args<T> a, March2Dets<T> b) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= b.first[b.count]) return;
#pragma unroll
  for (int d = 0; d < MARCH2_MAX_DETS; ++d)
    if (d < b.count && t >= b.first[d] && t < b.first[d + 1]) {
      const int64_t u = t - b.first[d];
      const int64_t cell = b.cell[d][u], plane = (int64_t)a.ny * a.nz;
      const int i = (int)(cell / plane), j = (int)((cell % plane) / a.nz), k = (int)(cell % a.nz);
      const bool in = i >= a.box[0] && i < a.box[1] && j >= a.box[2] && j < a.box[3] && k >= a.box[4] && k < a.box[5];
      const bool holed = i >= a.hole[0] && i < a.hole[1] && j >= a.hole[2] && j < a.hole[3] &&
                         k >= a.hole[4] && k < a.hole[5];
      if (!in || holed) continue;
      T e[3], ex[3] = {T(0), T(0), T(0)}, ey[3] = {T(0), T(0), T(0)}, ez[3] = {T(0), T(0), T(0)};
      march2_plain_E1<T>(a, i, j, k, e);
      const bool ox = i < a.nx - 1, oy = j < a.ny - 1, oz = k < a.nz - 1;
      if (ox) march2_plain_E1<T>(a, i + 1, j, k, ex);
      if (oy) march2_plain_E1<T>(a, i, j + 1, k, ey);
      if (oz) march2_plain_E1<T>(a, i, j, k + 1, ez);
      const T x_y = ox ? ex[1] - e[1] : T(0), x_z = ox ? ex[2] - e[2] : T(0);
      const T y_x = oy ? ey[0] - e[0] : T(0), y_z = oy ? ey[2] - e[2] : T(0);
      const T z_x = oz ? ez[0] - e[0] : T(0), z_y = oz ? ez[1] - e[1] : T(0);
      const T* h0 = a.H0 + cell * 3;
      T* oe = b.outE[d] + u * 3;
      T* oh = b.outH[d] + u * 3;
      oe[0] = e[0]; oe[1] = e[1]; oe[2] = e[2];
      oh[0] = h0[0] - a.ch[0] * (y_z - z_y);
      oh[1] = h0[1] - a.ch[1] * (z_x - x_z);
      oh[2] = h0[2] - a.ch[2] * (x_y - y_x);
    }
}

}  // namespace fdtdhip
