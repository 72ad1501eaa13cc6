// Large-grid Yee half-step kernels: one lane per cell, a wavefront per z-row
// segment.
//
// Workgroup = 256 lanes = 4 wavefronts; wavefront w handles 64 adjacent cells of
// row j = 4*blockIdx.y + w, plane i = x0 + blockIdx.z, starting at z = 64*blockIdx.x.
// A wave therefore reads 64 adjacent cells (768 contiguous bytes in float32) with
// every load, and i, j are wave-uniform (scalar registers, no index division).
//
// All loads of a lane are unconditional and issued back to back -- at a domain
// edge the neighbour address is clamped to the cell itself and the difference is
// discarded -- so the wave waits once for memory instead of once per neighbour.
// Neighbour re-reads (row j-1, plane i-1, cell k-1) are served by L1/L2/Infinity
// Cache; HBM sees each array once per half step.
// Per-cell arithmetic is shared with the shape-agnostic kernels (finish_E/H).
#pragma once

#include "plan.hpp"
#include "yee_generic.hpp"

namespace fdtdhip {

template <typename T, bool NT>
__device__ __forceinline__ void load3(const T* p, T& x, T& y, T& z) {
  if (NT) {
    x = __builtin_nontemporal_load(p);
    y = __builtin_nontemporal_load(p + 1);
    z = __builtin_nontemporal_load(p + 2);
  } else {
    x = p[0]; y = p[1]; z = p[2];
  }
}

template <typename T, bool NT>
__device__ __forceinline__ void store3(T* p, T x, T y, T z) {
  if (NT) {
    __builtin_nontemporal_store(x, p);
    __builtin_nontemporal_store(y, p + 1);
    __builtin_nontemporal_store(z, p + 2);
  } else {
    p[0] = x; p[1] = y; p[2] = z;
  }
}

// float64 result store through LDS.  A lane's cell is 24 B, which the compiler stores as a
// 16-byte and an 8-byte instruction: every 64-byte burst of the output then arrives in two
// partial pieces.  When all 64 lanes of the wave hold consecutive cells (a complete row
// segment), the wave instead lays its 1536 bytes out in LDS and writes them back as
// contiguous 16-byte pieces (lane l: piece l, and piece 64+l for l < 32): full lines.
template <bool NTS>
__device__ __forceinline__ void store3_wave(double* cell_ptr, double x, double y, double z,
                                            double* lds_wave) {
  // (the lanes of a wave always hold consecutive cells of one plane, in both lane mappings)
  const int lane = threadIdx.x & 63;
  const bool full = __builtin_amdgcn_read_exec() == ~0ull;
  if (!full) {
    store3<double, NTS>(cell_ptr, x, y, z);
    return;
  }
  double* l = lds_wave;
  l[lane * 3] = x; l[lane * 3 + 1] = y; l[lane * 3 + 2] = z;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  double* out = cell_ptr - lane * 3;   // the wave's first cell
  const double a0 = l[lane], a1 = l[64 + lane], a2 = l[128 + lane];
  if (NTS) {
    __builtin_nontemporal_store(a0, out + lane);
    __builtin_nontemporal_store(a1, out + 64 + lane);
    __builtin_nontemporal_store(a2, out + 128 + lane);
  } else {
    out[lane] = a0; out[64 + lane] = a1; out[128 + lane] = a2;
  }
}
template <bool NTS>
__device__ __forceinline__ void store3_wave(float* cell_ptr, float x, float y, float z, float*) {
  store3<float, NTS>(cell_ptr, x, y, z);   // 12-byte cells already store as full lines
}

// float64 loads through LDS.  A lane's 24-byte cell loads as a 16-byte and an 8-byte
// instruction at a 24-byte lane stride: every line is touched twice, in partial pieces.
// When the wave holds 64 consecutive cells (1536 contiguous bytes per array) the field being
// advanced (streamed once: non-temporal) and the wave's own segment of the driving field
// are loaded instead as dense 8-byte accesses -- lane l takes doubles l, 64+l and 128+l of
// the segment -- and handed to their cells through LDS.  The own segment in LDS also serves
// the z neighbour (the cell before the segment for the backward differences of the E half
// step, the cell after it for the forward differences of the H half step, fetched by three
// lanes); the x and y neighbours are read from memory, only the two components each
// difference needs.  MI355X, 512^3, vacuum, constants: 1.71 against 1.87 ms per launch
// (5.64 against 5.15 TB/s).
constexpr int DENSE_STAGE = 200 + 192;   // doubles per wave: 3 + 192 + 3 (+2 pad), then 192
// own[3], zn[2] (components 0, 1 of the z neighbour), f[3] of this lane's cell.
// gcell / fcell: this lane's cell in the driving / advanced field; first / last: element
// offsets of the arrays' first and one-past-last element relative to gcell's segment start.
template <bool FORWARD>
__device__ __forceinline__ void load_dense(const double* gcell, const double* fcell,
                                           int64_t seg_elem, int64_t total_elem, double* L,
                                           double own[3], double zn[2], double f[3]) {
  const int lane = threadIdx.x & 63;
  const double* gseg = gcell - 3 * lane;
  const double* fseg = fcell - 3 * lane;
  double* o = L + 3;
  double* ff = L + 200;
  const double g0 = gseg[lane], g1 = gseg[64 + lane], g2 = gseg[128 + lane];
  const double f0 = __builtin_nontemporal_load(fseg + lane);
  const double f1 = __builtin_nontemporal_load(fseg + 64 + lane);
  const double f2 = __builtin_nontemporal_load(fseg + 128 + lane);
  double edge = 0;
  if (lane < 3) {
    if (FORWARD) { if (seg_elem + 195 <= total_elem) edge = gseg[192 + lane]; }
    else if (seg_elem >= 3) edge = gseg[lane - 3];
  }
  o[lane] = g0; o[64 + lane] = g1; o[128 + lane] = g2;
  ff[lane] = f0; ff[64 + lane] = f1; ff[128 + lane] = f2;
  if (lane < 3) o[FORWARD ? 192 + lane : lane - 3] = edge;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  own[0] = o[3 * lane]; own[1] = o[3 * lane + 1]; own[2] = o[3 * lane + 2];
  const int z = FORWARD ? 3 * lane + 3 : 3 * lane - 3;
  zn[0] = o[z]; zn[1] = o[z + 1];
  f[0] = ff[3 * lane]; f[1] = ff[3 * lane + 1]; f[2] = ff[3 * lane + 2];
  __builtin_amdgcn_wave_barrier();   // the store staging below reuses ff
}

// Cell of this lane.  FLAT: 256 consecutive cells of the flattened (y, z) plane per
// workgroup (grid = blocks per plane x planes); otherwise a wave per z-row segment
// (grid = z tiles x row quads x planes).
template <bool FLAT>
__device__ __forceinline__ bool lane_cell(int x0, int ny, int nz, int rot_shift, int& i, int& j,
                                          int& k) {
  if (FLAT) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    i = x0 + blockIdx.y;
    j = r / nz;
    k = r - j * nz;
    return j < ny;
  }
  // Workgroups are dealt to the 8 XCDs round robin along blockIdx.x first.  When the
  // number of z segments is a multiple of 8 an unrotated mapping would give each XCD one
  // fixed z segment: good for reuse (x and y neighbours were fetched by the same XCD's
  // L2) but every cell of the z-low and z-high PML faces would land on two XCDs.
  // Rotating the segment by one every 16 block rows keeps nearly all of the reuse and
  // spreads the faces.
  k = ((blockIdx.x + (blockIdx.y >> rot_shift)) % gridDim.x) * 64 + (threadIdx.x & 63);
  j = blockIdx.y * 4 + (threadIdx.x >> 6);
  i = x0 + blockIdx.z;
  return k < nz && j < ny;
}

// (8 waves per SIMD: left to itself the compiler gives the variant with PMLs and objects 68
// vector registers, one wave fewer, which cost more than the object arithmetic itself)
template <typename T, bool HAS_PML, bool HAS_OBJ, bool FLAT, bool NT, bool NTS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void bulk_E_rows(BulkArgs<T> a) {
  int i, j, k;
  if (!lane_cell<FLAT>(a.x0, a.ny, a.nz, a.rot_shift, i, j, k)) return;
  const int64_t sy = 3 * (int64_t)a.nz, sx = sy * a.ny;
  const int64_t r3 = (int64_t)j * sy + 3 * k;
  const int64_t c3 = (int64_t)i * sx + r3;
  const T* __restrict__ g = a.G + c3;

  // plane 0 of an x-slab takes the x neighbour's two components from the ghost plane
  // (wave-uniform: the plane index is the same for the whole workgroup)
  const bool from_ghost = i == 0 && a.ghost != nullptr;
  const bool hasx = i > 0 || from_ghost, hasy = j > 0, hasz = k > 0;
  const T* gx = i > 0 ? g - sx : g;
  const T* gy = hasy ? g - sy : g;
  const T* gz = hasz ? g - 3 : g;

  T hx = 0, hy = 0, hz = 0, xx, xy = 0, xz = 0, yx = 0, yy, yz = 0, zx = 0, zy = 0, zz_;
  __shared__ T stage[sizeof(T) == 8 ? 4 * DENSE_STAGE : 1];
  T* const wave_stage = stage + (sizeof(T) == 8 ? (threadIdx.x >> 6) * DENSE_STAGE : 0);
  T e[3], m[3];
  bool dense = false;
  if constexpr (sizeof(T) == 8) {
    dense = __builtin_amdgcn_read_exec() == ~0ull;   // 64 consecutive cells (a prefix otherwise)
    if (dense) {
      // (requested before the LDS round trip of load_dense, so that all of the lane's
      // loads are in flight together)
      xy = gx[1]; xz = gx[2];
      yx = gy[0]; yz = gy[2];
      double own[3], zn[2];
      load_dense<false>(g, a.F + c3, c3 - 3 * (threadIdx.x & 63), 3 * (int64_t)a.nx * a.ny * a.nz,
                        wave_stage, own, zn, e);
      hx = own[0]; hy = own[1]; hz = own[2];
      zx = zn[0]; zy = zn[1];
    }
  }
  if (!dense) {
    load3<T, false>(g, hx, hy, hz);
    load3<T, false>(gx, xx, xy, xz);
    load3<T, false>(gy, yx, yy, yz);
    load3<T, false>(gz, zx, zy, zz_);
    load3<T, NT>(a.F + c3, e[0], e[1], e[2]);
  }
  cell_material<T, NT, HAS_OBJ>(a, c3, i, j, k, m);
  if (from_ghost) {
    const T* gh = a.ghost + ((int64_t)j * a.nz + k) * 2;
    xy = gh[0]; xz = gh[1];
  }
  (void)xx; (void)yy; (void)zz_;

  Diffs<T> d;
  d.x_y = hasx ? hy - xy : T(0);
  d.x_z = hasx ? hz - xz : T(0);
  d.y_x = hasy ? hx - yx : T(0);
  d.y_z = hasy ? hz - yz : T(0);
  d.z_x = hasz ? hx - zx : T(0);
  d.z_y = hasz ? hy - zy : T(0);
  const T hs[3] = {hx, hy, hz};
  finish_E<T, HAS_PML, HAS_OBJ>(a, i, j, k, d, e, m, hs);
  store3_wave<NTS>(a.F + c3, e[0], e[1], e[2], wave_stage + (sizeof(T) == 8 ? 200 : 0));
}

template <typename T, bool HAS_PML, bool FLAT, bool NT, bool NTS>
__global__ __launch_bounds__(256) void bulk_H_rows(BulkArgs<T> a) {
  int i, j, k;
  if (!lane_cell<FLAT>(a.x0, a.ny, a.nz, a.rot_shift, i, j, k)) return;
  const int64_t sy = 3 * (int64_t)a.nz, sx = sy * a.ny;
  const int64_t r3 = (int64_t)j * sy + 3 * k;
  const int64_t c3 = (int64_t)i * sx + r3;
  const T* __restrict__ g = a.G + c3;

  const bool from_ghost = i == a.nx - 1 && a.ghost != nullptr;   // last plane of an x-slab
  const bool hasx = i < a.nx - 1 || from_ghost, hasy = j < a.ny - 1, hasz = k < a.nz - 1;
  const T* gx = i < a.nx - 1 ? g + sx : g;
  const T* gy = hasy ? g + sy : g;
  const T* gz = hasz ? g + 3 : g;

  T ex = 0, ey = 0, ez = 0, xx, xy = 0, xz = 0, yx = 0, yy, yz = 0, zx = 0, zy = 0, zz_;
  __shared__ T stage[sizeof(T) == 8 ? 4 * DENSE_STAGE : 1];
  T* const wave_stage = stage + (sizeof(T) == 8 ? (threadIdx.x >> 6) * DENSE_STAGE : 0);
  T h[3], m[3];
  bool dense = false;
  if constexpr (sizeof(T) == 8) {
    dense = __builtin_amdgcn_read_exec() == ~0ull;
    if (dense) {
      xy = gx[1]; xz = gx[2];
      yx = gy[0]; yz = gy[2];
      double own[3], zn[2];
      load_dense<true>(g, a.F + c3, c3 - 3 * (threadIdx.x & 63), 3 * (int64_t)a.nx * a.ny * a.nz,
                       wave_stage, own, zn, h);
      ex = own[0]; ey = own[1]; ez = own[2];
      zx = zn[0]; zy = zn[1];
    }
  }
  if (!dense) {
    load3<T, false>(g, ex, ey, ez);
    load3<T, false>(gx, xx, xy, xz);
    load3<T, false>(gy, yx, yy, yz);
    load3<T, false>(gz, zx, zy, zz_);
    load3<T, NT>(a.F + c3, h[0], h[1], h[2]);
  }
  cell_material<T, NT, false>(a, c3, i, j, k, m);
  if (from_ghost) {
    const T* gh = a.ghost + ((int64_t)j * a.nz + k) * 2;
    xy = gh[0]; xz = gh[1];
  }
  (void)xx; (void)yy; (void)zz_;

  Diffs<T> d;
  d.x_y = hasx ? xy - ey : T(0);
  d.x_z = hasx ? xz - ez : T(0);
  d.y_x = hasy ? yx - ex : T(0);
  d.y_z = hasy ? yz - ez : T(0);
  d.z_x = hasz ? zx - ex : T(0);
  d.z_y = hasz ? zy - ey : T(0);
  finish_H<T, HAS_PML>(a, i, j, k, d, h, m);
  store3_wave<NTS>(a.F + c3, h[0], h[1], h[2], wave_stage + (sizeof(T) == 8 ? 200 : 0));
}

}  // namespace fdtdhip
