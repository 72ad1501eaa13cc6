// EXPERIMENT, NOT BUILT INTO THE LIBRARY: the marching whole-step kernel extended to plans
// with PML faces (step_march_pml below; host side in yee_march_pml_host.diff).  Bitwise equal
// to two half steps on every scene tried (fields and memory variables, six faces in the order
// DomainBorderPML registers them, unequal thicknesses, missing faces), but slower than the two
// half-step launches it was meant to replace.  MI355X, 512^3 float32, ms per step:
//
//                                         x faces   six faces   six faces + 2 boxes
//   two half-step launches                  1.71       1.77          1.85
//   faces evaluated row after row            --      4.6 .. 6.2       --
//   all memory variables of a plane
//     requested with the fields, 4 rows     2.50       2.68          3.91
//   the same, one plane ahead, 2 rows       2.60       3.18          4.14
//   (step_march, no faces: 1.25)
//
// Counters (rocprofv3, per wave and plane): step_march 549 VALU / 208 SALU instructions for
// nine row evaluations; the 4-row kernel below 1263 / 720, the 2-row one-plane-ahead form
// 720 / 452 for five.  The face code is branch-free for a wave that meets no face, but its
// descriptors, offsets and conditions are wave-uniform values that do not fit the scalar
// registers next to the kernel's own (106 of 106 used): 300 to 950 v_readlane / v_writelane
// per loop body, each a VALU instruction.  The kernel is then bound by instruction issue, not
// by memory.  A register allocated to an unused lane of a wider load, and registers of stores
// in flight reused by the first writes of the next plane, each cost a full
// `s_waitcnt vmcnt(0)` per plane on top (found in the ISA; fixed below where noted).
// One compiler trap worth keeping: __builtin_bit_cast(float, v.y) on an element of a vector
// returned by a buffer-load builtin reads element 0 (hipcc 7.2); __uint_as_float is right.
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
// MI355X, 512^3 float32 vacuum, constants: 1.17 ms per step against 1.66 ms for two
// half-step launches (TY = 4: 97 registers, 4 waves per SIMD; TY = 2: 1.26 ms, 3: 1.23,
// 5: 1.32, 8: 1.40; chunks of 16 to 128 planes within 2 %, a single chunk 2.6 ms).
#pragma once

#include "plan.hpp"
#include "yee_generic.hpp"
#include "yee_step.hpp"

namespace fdtdhip {

constexpr int MARCH_MAX_WAVES = 16;   // z rows of up to 1024 cells
constexpr int MARCH_MAX_LX = 128;     // planes per chunk at most

// One PML face for both half steps of the step.  psi arrays: (2, slab cells), the slab being
// the grid with its extent along the normal axis replaced by t (reference
// fdtd/boundaries.py:296-343).
template <typename T>
struct MarchFace {
  const T* psiE;   // memory variables of the E half step before the step ...
  T* psiE_out;     // ... and where the advanced ones go (the second buffer: rows evaluated
                   // redundantly by the neighbour tile still read the old ones)
  T* psiH;         // memory variables of the H half step, advanced in place (owners only)
  const T* bE; const T* cE; const T* bH; const T* cH;   // (t) profiles along the normal axis
  int lo, t;       // first grid index along the normal axis and thickness; t == 0: no face
};

template <typename T>
struct MarchArgs {
  StepArgs<T> s;
  int lx;          // planes per chunk
  int band;        // tiles per XCD (grid x = 8 * band), or 0: tile = blockIdx.x
  // PML kernels: the faces by normal axis and side (a cell lies in at most one face per axis)
  // and the order in which the axes' corrections are added to a cell (registration order of
  // the faces, which the sum of two corrections to one component depends on bit for bit).
  MarchFace<T> face[3][2];
  int order[3];
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
          En[r][0] = __builtin_nontemporal_load(E0 + c3);
          En[r][1] = __builtin_nontemporal_load(E0 + c3 + 1);
          En[r][2] = __builtin_nontemporal_load(E0 + c3 + 2);
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
          __builtin_nontemporal_store(En[r][0], o);
          __builtin_nontemporal_store(En[r][1], o + 1);
          __builtin_nontemporal_store(En[r][2], o + 2);
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
          __builtin_nontemporal_store(h[0], o);
          __builtin_nontemporal_store(h[1], o + 1);
          __builtin_nontemporal_store(h[2], o + 2);
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

__device__ __forceinline__ float wave_uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ double wave_uniform(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)(b & 0xffffffffll));
  const unsigned hi = __builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// The side (0, 1) of the face of one axis that holds `coord`, or -1.
template <typename T>
__device__ __forceinline__ int face_at(const MarchFace<T> (&f)[2], int coord) {
  if (f[0].t > 0 && coord >= f[0].lo && coord < f[0].lo + f[0].t) return 0;
  if (f[1].t > 0 && coord >= f[1].lo && coord < f[1].lo + f[1].t) return 1;
  return -1;
}
// The side whose slab meets [c0, c1], or -1 (the plan guarantees that not both do).
template <typename T>
__device__ __forceinline__ int face_near(const MarchFace<T> (&f)[2], int c0, int c1) {
  if (f[0].t > 0 && f[0].lo <= c1 && f[0].lo + f[0].t > c0) return 0;
  if (f[1].t > 0 && f[1].lo <= c1 && f[1].lo + f[1].t > c0) return 1;
  return -1;
}

// One face acting on one cell: the arithmetic of pml_face (yee_generic.hpp; reference
// fdtd/boundaries.py:398-476) on memory variables q0, q1 that were loaded earlier.
// drive: the slab-local index takes the field difference (E half step: >= 1, H: <= t-2).
template <typename T, int AXIS, bool PHASE_E>
__device__ __forceinline__ void march_face(T q0, T q1, T b, T c, bool drive, const Diffs<T>& d,
                                           T sc, const T m[3], T f[3], T out[2]) {
  constexpr int u = (AXIS + 1) % 3, v = (AXIS + 2) % 3;
  const T dFu = AXIS == 0 ? d.x_y : (AXIS == 1 ? d.y_z : d.z_x);
  const T dFv = AXIS == 0 ? d.x_z : (AXIS == 1 ? d.y_x : d.z_y);
  T p0 = q0 * b, p1 = q1 * b;
  if (drive) { p0 += dFv * c; p1 += dFu * c; }
  out[0] = p0; out[1] = p1;
  if (PHASE_E) {
    f[u] += (sc * m[u]) * (-p0);
    f[v] += (sc * m[v]) * p1;
  } else {
    f[u] -= (sc * m[u]) * (-p0);
    f[v] -= (sc * m[v]) * p1;
  }
}

// A descriptor chosen between the two sides of an axis, member by member (scalar selects: an
// index into the kernel argument that is only known at run time would move the whole
// argument into scratch memory).  side < 0: no face (t = 0).
template <typename T>
__device__ __forceinline__ MarchFace<T> pick_face(const MarchFace<T> (&f)[2], int side) {
  MarchFace<T> r;
  const bool hi = side == 1;
  r.psiE = hi ? f[1].psiE : f[0].psiE;
  r.psiE_out = hi ? f[1].psiE_out : f[0].psiE_out;
  r.psiH = hi ? f[1].psiH : f[0].psiH;
  r.bE = hi ? f[1].bE : f[0].bE;
  r.cE = hi ? f[1].cE : f[0].cE;
  r.bH = hi ? f[1].bH : f[0].bH;
  r.cH = hi ? f[1].cH : f[0].cH;
  r.lo = hi ? f[1].lo : f[0].lo;
  r.t = side < 0 ? 0 : (hi ? f[1].t : f[0].t);
  return r;
}

// Buffer addressing for the kernel with PML faces: a 128-bit descriptor in scalar registers,
// a wave-uniform 32-bit byte offset in a scalar register (row, plane) and a 32-bit lane
// offset in one vector register -- four lane offsets serve every access of the kernel, where
// 64-bit lane addresses per access kept a hundred registers busy.  No range check is wanted
// (all ones in the descriptor's size word): lanes outside the grid are masked by the caller.
using BufferRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BufferRsrc buffer_of(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
typedef unsigned int march_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int march_u3 __attribute__((ext_vector_type(3)));
typedef unsigned int march_u4 __attribute__((ext_vector_type(4)));

// NT: non-temporal (streamed once).
template <bool NT>
__device__ __forceinline__ void buf_load3(BufferRsrc r, unsigned lane, unsigned row, float v[3]) {
  const march_u3 x = __builtin_amdgcn_raw_buffer_load_b96(r, lane, row, NT ? 2 : 0);
  // (__builtin_bit_cast applied to an element of the vector reads element 0 whatever the index)
  v[0] = __uint_as_float(x.x); v[1] = __uint_as_float(x.y); v[2] = __uint_as_float(x.z);
}
template <bool NT>
__device__ __forceinline__ void buf_load3(BufferRsrc r, unsigned lane, unsigned row, double v[3]) {
  const march_u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, lane, row, NT ? 2 : 0);
  const march_u2 y = __builtin_amdgcn_raw_buffer_load_b64(r, lane + 16u, row, NT ? 2 : 0);
  v[0] = __builtin_bit_cast(double, march_u2{x.x, x.y});
  v[1] = __builtin_bit_cast(double, march_u2{x.z, x.w});
  v[2] = __builtin_bit_cast(double, y);
}
__device__ __forceinline__ void buf_load2(BufferRsrc r, unsigned lane, unsigned row, float& a, float& b) {
  const march_u2 x = __builtin_amdgcn_raw_buffer_load_b64(r, lane, row, 0);
  a = __uint_as_float(x.x); b = __uint_as_float(x.y);
}
__device__ __forceinline__ void buf_load2(BufferRsrc r, unsigned lane, unsigned row, double& a, double& b) {
  const march_u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, lane, row, 0);
  a = __builtin_bit_cast(double, march_u2{x.x, x.y});
  b = __builtin_bit_cast(double, march_u2{x.z, x.w});
}
__device__ __forceinline__ float buf_load(BufferRsrc r, unsigned lane, unsigned row, float) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane, row, 0));
}
__device__ __forceinline__ double buf_load(BufferRsrc r, unsigned lane, unsigned row, double) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, lane, row, 0));
}
__device__ __forceinline__ void buf_store(BufferRsrc r, unsigned lane, unsigned row, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, lane, row, 0);
}
__device__ __forceinline__ void buf_store(BufferRsrc r, unsigned lane, unsigned row, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(march_u2, v), r, lane, row, 0);
}
__device__ __forceinline__ void buf_store3_nt(BufferRsrc r, unsigned lane, unsigned row, const float v[3]) {
  const march_u3 x = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]),
                      __builtin_bit_cast(unsigned, v[2])};
  __builtin_amdgcn_raw_buffer_store_b96(x, r, lane, row, 2);
}
__device__ __forceinline__ void buf_store3_nt(BufferRsrc r, unsigned lane, unsigned row, const double v[3]) {
  const march_u2 a = __builtin_bit_cast(march_u2, v[0]), b = __builtin_bit_cast(march_u2, v[1]);
  __builtin_amdgcn_raw_buffer_store_b128(march_u4{a.x, a.y, b.x, b.y}, r, lane, row, 2);
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(march_u2, v[2]), r, lane + 16u, row, 2);
}

// Everything a wave reads from memory for one iteration of the marching step: the fields of
// plane i (rows of the tile, the row below them, the row above for its x and z components, the
// cells before along z), the memory variables of the E half step at plane i and of the H half
// step at plane i-1, and the x profile values of both.
template <typename T, int TY>
struct PlaneLoads {
  T H[TY + 1][3], E[TY + 1][3], zx[TY + 1], zy[TY + 1], ux, uz;
  T qxe[TY + 1][2], qye[TY + 1][2], qze[TY + 1][2], qxh[TY][2], qyh[TY][2], qzh[TY][2];
  T xbE, xcE, xbH, xcH, ybE[TY + 1], ycE[TY + 1], ybH[TY], ycH[TY];
};

// The marching step for plans with PML faces and/or objects.  As step_march, and:
//   * everything plane i needs from memory -- fields and the memory variables of its faces --
//     is requested one iteration ahead, while plane i-1 is being evaluated, into registers of
//     its own (PlaneLoads).  A workgroup then hides its memory latency itself instead of
//     relying on a second workgroup on the CU, which the registers of these variants do not
//     leave room for; and the registers a store still reads are not the ones the next loads
//     land in (with everything of a plane requested and used inside one iteration the
//     compiler's register allocation forced a wait for the previous plane's stores before the
//     first load of the next: 2.5 ms per step at 512^3 float32 where step_march takes 1.25).
//     (Evaluated row after row like the half-step kernels do, the face code made every
//     workgroup wait, plane by plane, for a chain of round trips per row: 4.6 ms.)
//   * buffer addressing (above), so that the addresses cost four vector registers in all.
// The plan guarantees that a chunk of planes, a tile of rows and a wave's z segment each meet
// at most one face of their axis, so a wave works with one descriptor per axis; that one plane
// of a field and one array of memory variables hold fewer than 2^31 bytes (32-bit offsets).
template <typename T, int TY, bool HAS_OBJ, bool CONST, int MW = MARCH_MAX_WAVES>
__global__ __launch_bounds__(64 * MW) void step_march_pml(MarchArgs<T> ma) {
  const StepArgs<T>& a = ma.s;
  const BulkArgs<T>& A = a.e;
  constexpr unsigned ES = (unsigned)sizeof(T);
  // (the wave index read through one lane: a scalar to the compiler, and with it everything
  // a wave selects by its z segment)
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int k = w * 64 + lane;
  const int nx = A.nx, ny = A.ny, nz = A.nz;
  const bool kin = k < nz;
  const int tile = ma.band > 0 ? (int)(blockIdx.x & 7) * ma.band + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile * TY >= ny) return;
  const int j0 = tile * TY;
  const int i0 = blockIdx.y * ma.lx, i1 = min(i0 + ma.lx, nx);
  const int64_t sy = 3 * (int64_t)nz, sx = sy * ny;
  const unsigned rowb = 3u * (unsigned)nz * ES;     // bytes of one field row
  // lane offsets in bytes: the cell in a field row, the cell before it along z (the cell
  // itself at k = 0), the cell in a row of x- or y-face memory variables, in a z-face row
  const unsigned kb3 = (unsigned)k * 3u * ES;
  const unsigned kzb = k > 0 ? kb3 - 3u * ES : kb3;
  const unsigned kb = (unsigned)k * ES;
  unsigned lzb = 0;
  // new E (components x, y) of each wave's first lane, for the last lane of the wave before
  __shared__ T xch[2][MARCH_MAX_WAVES + 1][TY][2];
  T Hp[TY + 1][3], Ep[TY + 1][3];
#pragma unroll
  for (int r = 0; r <= TY; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) { Hp[r][c] = T(0); Ep[r][c] = T(0); }
  if (i0 > 0 && kin) {
    const BufferRsrc hb = buffer_of(A.G + (int64_t)(i0 - 1) * sx);
#pragma unroll
    for (int r = 0; r <= TY; ++r)
      if (j0 + r < ny) buf_load3<false>(hb, kb3, (unsigned)(j0 + r) * rowb, Hp[r]);
  }
  const bool hasz = k > 0, hasz1 = k < nz - 1;

  // The PML face of each axis this workgroup's planes (x), rows (y) and this wave's cells (z)
  // can meet.  Only its side, first index and thickness stay in registers: the pointers are
  // read from the kernel argument where a face acts (few waves, few planes), so that the
  // waves and planes without faces do not pay for them in scalar registers.
  const int xs = face_near<T>(ma.face[0], i0, i1), ys = face_near<T>(ma.face[1], j0, j0 + TY),
            zs = face_near<T>(ma.face[2], 64 * w, 64 * w + 63);
  const int xlo = xs == 1 ? ma.face[0][1].lo : ma.face[0][0].lo, xt = xs < 0 ? 0 : (xs == 1 ? ma.face[0][1].t : ma.face[0][0].t);
  const int ylo = ys == 1 ? ma.face[1][1].lo : ma.face[1][0].lo, yt = ys < 0 ? 0 : (ys == 1 ? ma.face[1][1].t : ma.face[1][0].t);
  const int zlo = zs == 1 ? ma.face[2][1].lo : ma.face[2][0].lo, zt = zs < 0 ? 0 : (zs == 1 ? ma.face[2][1].t : ma.face[2][0].t);
#define FACE_MEMBER(ax, side, member) ((side) == 1 ? ma.face[ax][1].member : ma.face[ax][0].member)
  bool yin[TY + 1];
#pragma unroll
  for (int r = 0; r <= TY; ++r) {
    const int l = j0 + r - ylo;
    yin[r] = l >= 0 && l < yt && j0 + r < ny;
  }
  const int lz = k - zlo;
  const bool zin = kin && lz >= 0 && lz < zt;
  T zbE = T(0), zcE = T(0), zbH = T(0), zcH = T(0);
  if (zin) {
    lzb = (unsigned)lz * ES;
    zbE = FACE_MEMBER(2, zs, bE)[lz]; zcE = FACE_MEMBER(2, zs, cE)[lz];
    zbH = FACE_MEMBER(2, zs, bH)[lz]; zcH = FACE_MEMBER(2, zs, cH)[lz];
  }
  // second component of an array of memory variables: one slab further
  const unsigned xslab = (unsigned)xt * (unsigned)ny * (unsigned)nz * ES,
                 yslab = (unsigned)nx * (unsigned)yt * (unsigned)nz * ES,
                 zslab = (unsigned)nx * (unsigned)ny * (unsigned)zt * ES;
  const unsigned prow = (unsigned)nz * ES, zrow = (unsigned)zt * ES;   // bytes of a row of them
  // byte offsets of the memory variables of row 0 of the tile at plane p, per axis
  auto ox = [&](int p) { return ((unsigned)(p - xlo) * (unsigned)ny + (unsigned)j0) * prow; };
  auto oy = [&](int p) { return ((unsigned)p * (unsigned)yt + (unsigned)(j0 - ylo)) * prow; };
  auto oz = [&](int p) { return ((unsigned)p * (unsigned)ny + (unsigned)j0) * zrow; };
  auto in_x = [&](int p) { return p >= xlo && p < xlo + xt; };

  // Requests what iteration i needs (the E half step of plane i, the H half step of plane i-1).
  auto fetch = [&](int i, PlaneLoads<T, TY>& L) {
    if (!kin) return;
    const int ih = i - 1;
    if (ih >= i0) {
      const bool xh = in_x(ih);
#pragma unroll
      for (int r = 0; r < TY; ++r) {
        if (j0 + r >= ny) continue;
        if (xh) {
          const BufferRsrc b = buffer_of(FACE_MEMBER(0, xs, psiH));
          L.qxh[r][0] = buf_load(b, kb, ox(ih) + r * prow, T());
          L.qxh[r][1] = buf_load(b, kb, ox(ih) + r * prow + xslab, T());
        }
        if (yin[r]) {
          const BufferRsrc b = buffer_of(FACE_MEMBER(1, ys, psiH));
          L.qyh[r][0] = buf_load(b, kb, oy(ih) + r * prow, T());
          L.qyh[r][1] = buf_load(b, kb, oy(ih) + r * prow + yslab, T());
        }
        if (zin) {
          const BufferRsrc b = buffer_of(FACE_MEMBER(2, zs, psiH));
          L.qzh[r][0] = buf_load(b, lzb, oz(ih) + r * zrow, T());
          L.qzh[r][1] = buf_load(b, lzb, oz(ih) + r * zrow + zslab, T());
        }
      }
      if (xh) { L.xbH = FACE_MEMBER(0, xs, bH)[ih - xlo]; L.xcH = FACE_MEMBER(0, xs, cH)[ih - xlo]; }
#pragma unroll
      for (int r = 0; r < TY; ++r)
        if (yin[r]) { L.ybH[r] = FACE_MEMBER(1, ys, bH)[j0 + r - ylo]; L.ycH[r] = FACE_MEMBER(1, ys, cH)[j0 + r - ylo]; }
    }
    if (i < nx) {
      const bool xe = in_x(i);
      const BufferRsrc hb = buffer_of(A.G + (int64_t)i * sx), eb = buffer_of(A.F + (int64_t)i * sx);
#pragma unroll
      for (int r = 0; r <= TY; ++r) {
        if (j0 + r >= ny) continue;
        const unsigned row = (unsigned)(j0 + r) * rowb;
        buf_load3<false>(hb, kb3, row, L.H[r]);
        buf_load3<true>(eb, kb3, row, L.E[r]);
        buf_load2(hb, kzb, row, L.zx[r], L.zy[r]);
        if (xe) {
          const BufferRsrc b = buffer_of(FACE_MEMBER(0, xs, psiE));
          L.qxe[r][0] = buf_load(b, kb, ox(i) + r * prow, T());
          L.qxe[r][1] = buf_load(b, kb, ox(i) + r * prow + xslab, T());
        }
        if (yin[r]) {
          const BufferRsrc b = buffer_of(FACE_MEMBER(1, ys, psiE));
          L.qye[r][0] = buf_load(b, kb, oy(i) + r * prow, T());
          L.qye[r][1] = buf_load(b, kb, oy(i) + r * prow + yslab, T());
        }
        if (zin) {
          const BufferRsrc b = buffer_of(FACE_MEMBER(2, zs, psiE));
          L.qze[r][0] = buf_load(b, lzb, oz(i) + r * zrow, T());
          L.qze[r][1] = buf_load(b, lzb, oz(i) + r * zrow + zslab, T());
        }
      }
      if (j0 > 0) {   // (two single loads: the y component is not needed, and a register of a
                      // wider load that nothing reads is handed to another value while the
                      // load is still in flight)
        L.ux = buf_load(hb, kb3, (unsigned)(j0 - 1) * rowb, T());
        L.uz = buf_load(hb, kb3 + 2u * ES, (unsigned)(j0 - 1) * rowb, T());
      }
      if (xe) { L.xbE = FACE_MEMBER(0, xs, bE)[i - xlo]; L.xcE = FACE_MEMBER(0, xs, cE)[i - xlo]; }
#pragma unroll
      for (int r = 0; r <= TY; ++r)
        if (yin[r]) { L.ybE[r] = FACE_MEMBER(1, ys, bE)[j0 + r - ylo]; L.ycE[r] = FACE_MEMBER(1, ys, cE)[j0 + r - ylo]; }
    }
  };

  PlaneLoads<T, TY> next;
  fetch(i0, next);
  for (int i = i0; i <= i1; ++i) {
    // what was requested an iteration ago; then the requests for the next iteration
    const PlaneLoads<T, TY> L = next;
    if (i < i1) fetch(i + 1, next);
    T En[TY + 1][3];
    const bool plane = i < nx;
    const int ih = i - 1;
    const bool xe = plane && in_x(i), xh = i > i0 && in_x(ih);
    if (plane && kin) {
      const bool hasx = i > 0;
      const BufferRsrc e1 = buffer_of(a.E1 + (int64_t)i * sx);
#pragma unroll
      for (int r = 0; r <= TY; ++r) {
        const int j = j0 + r;
        if (j >= ny) continue;
        En[r][0] = L.E[r][0]; En[r][1] = L.E[r][1]; En[r][2] = L.E[r][2];
        const int64_t c3 = (int64_t)i * sx + (int64_t)j * sy + 3 * k;
        const bool hasy = j > 0;
        const T yx = r == 0 ? L.ux : L.H[r == 0 ? 0 : r - 1][0];
        const T yz = r == 0 ? L.uz : L.H[r == 0 ? 0 : r - 1][2];
        Diffs<T> d;
        d.x_y = hasx ? L.H[r][1] - Hp[r][1] : T(0);
        d.x_z = hasx ? L.H[r][2] - Hp[r][2] : T(0);
        d.y_x = hasy ? L.H[r][0] - yx : T(0);
        d.y_z = hasy ? L.H[r][2] - yz : T(0);
        d.z_x = hasz ? L.H[r][0] - L.zx[r] : T(0);
        d.z_y = hasz ? L.H[r][1] - L.zy[r] : T(0);
        const bool owner = r < TY && i < i1;
        T m[3];
        if (CONST && !HAS_OBJ) { m[0] = A.bg[0]; m[1] = A.bg[1]; m[2] = A.bg[2]; }
        else cell_material<T, false, HAS_OBJ>(A, c3, i, j, k, m);
        finish_E<T, false, HAS_OBJ>(A, i, j, k, d, En[r], m, L.H[r], owner);
#pragma nounroll
        for (int n = 0; n < 3; ++n) {
          const int ax = ma.order[n];
          T out[2];
          if (ax == 0) {
            if (xe) {
              march_face<T, 0, true>(L.qxe[r][0], L.qxe[r][1], L.xbE, L.xcE, i - xlo >= 1, d, A.sc, m, En[r], out);
              if (owner) {
                const BufferRsrc b = buffer_of(FACE_MEMBER(0, xs, psiE_out));
                buf_store(b, kb, ox(i) + r * prow, out[0]);
                buf_store(b, kb, ox(i) + r * prow + xslab, out[1]);
              }
            }
          } else if (ax == 1) {
            if (yin[r]) {
              march_face<T, 1, true>(L.qye[r][0], L.qye[r][1], L.ybE[r], L.ycE[r], j - ylo >= 1, d, A.sc, m, En[r], out);
              if (owner) {
                const BufferRsrc b = buffer_of(FACE_MEMBER(1, ys, psiE_out));
                buf_store(b, kb, oy(i) + r * prow, out[0]);
                buf_store(b, kb, oy(i) + r * prow + yslab, out[1]);
              }
            }
          } else if (zin) {
            march_face<T, 2, true>(L.qze[r][0], L.qze[r][1], zbE, zcE, lz >= 1, d, A.sc, m, En[r], out);
            if (owner) {
              const BufferRsrc b = buffer_of(FACE_MEMBER(2, zs, psiE_out));
              buf_store(b, lzb, oz(i) + r * zrow, out[0]);
              buf_store(b, lzb, oz(i) + r * zrow + zslab, out[1]);
            }
          }
        }
        if (a.nsrc > 0) march_sources<T>(a, c3, En[r]);
        if (owner) buf_store3_nt(e1, kb3, (unsigned)j * rowb, En[r]);
      }
    }
    if (i > i0) {
      // new H of plane i-1, owned rows
      const BufferRsrc h1 = buffer_of(a.H1 + (int64_t)ih * sx);
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
          finish_H<T, false>(a.h, ih, j, k, d, h, m);
#pragma nounroll
          for (int n = 0; n < 3; ++n) {
            const int axn = ma.order[n];
            T out[2];
            if (axn == 0) {
              if (xh) {
                march_face<T, 0, false>(L.qxh[r][0], L.qxh[r][1], L.xbH, L.xcH, ih - xlo <= xt - 2, d, a.h.sc, m, h, out);
                const BufferRsrc b = buffer_of(FACE_MEMBER(0, xs, psiH));
                buf_store(b, kb, ox(ih) + r * prow, out[0]);
                buf_store(b, kb, ox(ih) + r * prow + xslab, out[1]);
              }
            } else if (axn == 1) {
              if (yin[r]) {
                march_face<T, 1, false>(L.qyh[r][0], L.qyh[r][1], L.ybH[r], L.ycH[r], j - ylo <= yt - 2, d, a.h.sc, m, h, out);
                const BufferRsrc b = buffer_of(FACE_MEMBER(1, ys, psiH));
                buf_store(b, kb, oy(ih) + r * prow, out[0]);
                buf_store(b, kb, oy(ih) + r * prow + yslab, out[1]);
              }
            } else if (zin) {
              march_face<T, 2, false>(L.qzh[r][0], L.qzh[r][1], zbH, zcH, lz <= zt - 2, d, a.h.sc, m, h, out);
              const BufferRsrc b = buffer_of(FACE_MEMBER(2, zs, psiH));
              buf_store(b, lzb, oz(ih) + r * zrow, out[0]);
              buf_store(b, lzb, oz(ih) + r * zrow + zslab, out[1]);
            }
          }
          buf_store3_nt(h1, kb3, (unsigned)j * rowb, h);
        }
      }
    }
#pragma unroll
    for (int r = 0; r <= TY; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) { Hp[r][c] = L.H[r][c]; Ep[r][c] = En[r][c]; }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < TY; ++r) { xch[i & 1][w][r][0] = En[r][0]; xch[i & 1][w][r][1] = En[r][1]; }
    }
    __syncthreads();
  }
#undef FACE_MEMBER
}

}  // namespace fdtdhip
