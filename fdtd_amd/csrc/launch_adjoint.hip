// Reverse sweep of libfdtd_hip: the adjoint of the half steps (yee_adjoint.hpp) driven
// backwards through time, stands in for torch autograd over the reference's step loop
// (reference examples/learnable_focus.py:98-128).
#include "plan_host.hpp"
#include "yee_generic.hpp"
#include "yee_tiled.hpp"
#include "yee_adjoint.hpp"
#include "yee_adjoint_march.hpp"

namespace fdtdhip {

namespace {

// The adjoint of one half step is: reverse of the post pieces, pass 1 per cell
// (adjoint_cells), pass 2 gather (adjoint_gather).  On x-slabs the caller exchanges one
// plane of `work` with a neighbour rank between the two passes.
// Large planes with well-filled 64-lane row segments take the row lane mapping.
bool adjoint_rows(const fdtdhip_plan* p) {
  const int segs = (p->nz + 63) / 64;
  return (int64_t)p->ny * p->nz >= 4096 && p->nz >= 48 && p->nz * 10 >= segs * 64 * 9 &&
         p->nx <= 65535 && (p->ny + 3) / 4 <= 65535;
}

// FDTDHIP_ADJ_DIRECT: 0 = never derive difference adjoints from the field adjoint (plain two
// passes everywhere), 2 = wherever the plan allows, whatever the grid size (tests); default:
// on grids large enough for it to matter.
int adjoint_direct_mode() {
  static const int mode = [] {
    const char* e = getenv("FDTDHIP_ADJ_DIRECT");
    return e ? atoi(e) : 1;
  }();
  return mode;
}

// `flip`: 0 the memory-variable adjoints in place, 1 from the first to the second arrays, 2 back
template <typename T>
int adjoint_args(fdtdhip_plan* p, bool phaseE, T* lamE, T* lamH, T* work, AdjArgs<T>& a, int flip = 0) {
  a.lamF = phaseE ? lamE : lamH;
  a.lamG = phaseE ? lamH : lamE;
  a.mat = (const T*)(phaseE ? p->inv_eps : p->inv_mu);
  a.work = work;
  a.nx = p->nx; a.ny = p->ny; a.nz = p->nz;
  a.sc = (T)p->sc;
  a.npml = (int)p->pmls.size();
  a.nobj = phaseE ? (int)p->objs.size() : 0;
  a.obj = (const ObjK<T>*)p->obj_table;
  fill_object_override<T>(p, a);
  a.trace_step = phaseE ? p->trace_cursor : -1;
  for (const PmlH& h : p->pmls)
    if (!(phaseE ? h.lamE : h.lamH)) return fail(-1, "PML adjoint arrays not set");
  if (int rc = sync_pml_tables<T>(p)) return rc;
  a.pml = pml_table<T>(p, phaseE ? (flip == 0 ? TAB_ADJ_E : (flip == 1 ? TAB_ADJ_E_AB : TAB_ADJ_E_BA))
                                 : (flip == 0 ? TAB_ADJ_H : (flip == 1 ? TAB_ADJ_H_AB : TAB_ADJ_H_BA)));
  a.open_lo = p->H_below ? 1 : 0;
  a.open_hi = p->E_above ? 1 : 0;
  a.wghost = nullptr;
  const int w = phaseE ? 0 : 1;
  a.mat_const = p->mat_const[w] ? 1 : 0;
  for (int c = 0; c < 3; ++c) a.bg[c] = (T)p->mat_bg[w][c];
  BulkArgs<T> boxes;
  fill_core<T>(p, boxes);
  for (int n = 0; n < 6; ++n) { a.hull[n] = boxes.hull[n]; a.core[n] = boxes.core[n]; }
  for (int n = 0; n < MAX_PML; ++n) {
    a.face_axis[n] = boxes.face_axis[n];
    a.face_lo[n] = boxes.face_lo[n];
    a.face_hi[n] = boxes.face_hi[n];
  }
  a.rot_shift = boxes.rot_shift;
  // (pass 1 is then launched box by box with planes and row quads as grid dimensions)
  const int mode = adjoint_direct_mode();
  a.direct = mode > 0 && a.mat_const && p->nx <= 65535 &&
                     (p->ny + 3) / 4 <= 65535 &&
                     (mode > 1 || adjoint_rows(p) || p->cells() > (int64_t)(1 << 19)) ? 1 : 0;
  return 0;
}

// Launch table of a list of boxes (empty ones left out).
int cell_box_table(const std::vector<CellBox>& boxes, CellBoxes& tab) {
  tab.count = 0;
  tab.first[0] = 0;
  for (const CellBox& b : boxes) {
    if (b.i1 <= b.i0 || b.j1 <= b.j0 || b.k1 <= b.k0) continue;
    if (tab.count >= MAX_CELL_BOXES) return fail(-1, "internal: too many boxes for one launch");
    int gx, gy;
    box_grid(b, gx, gy);
    tab.box[tab.count] = b;
    tab.gx[tab.count] = gx;
    tab.gy[tab.count] = gy;
    tab.first[tab.count + 1] = tab.first[tab.count] + (int64_t)gx * gy * (b.i1 - b.i0);
    ++tab.count;
  }
  for (int m = tab.count; m < MAX_CELL_BOXES; ++m) {
    tab.box[m] = CellBox{0, 0, 0, 0, 0, 0};
    tab.gx[m] = tab.gy[m] = 1;
    tab.first[m + 1] = tab.first[tab.count];
  }
  if (tab.first[tab.count] >= (1ll << 31))
    return fail(-1, "internal: pass 1 of the adjoint over too many workgroups");
  return 0;
}

// The cells that keep the two-pass form when the plain ones do not (AdjArgs::direct): six
// slabs around the part of the PML-free core whose 3x3x3 blocks stay inside it ([s0, s1) per
// axis), and the hull of the objects grown by one cell where it lies in that part (`hole`,
// all zero without objects).  The boxes are disjoint: pass 1 has side effects
// (memory-variable adjoints, object gradients).
// boxes: [0] the slab below [s0, s1) along x, [1] the slab above it, [2..5] the y and z slabs
// between them, [6] (with objects) the box around the objects; empty ones included.
template <typename T>
void adjoint_shell_box_list(const fdtdhip_plan* p, const AdjArgs<T>& a, std::vector<CellBox>& boxes,
                            int (&s0)[3], int (&s1)[3], int (&hole)[6]) {
  const int n[3] = {p->nx, p->ny, p->nz};
  for (int d = 0; d < 3; ++d) {
    s0[d] = a.core[2 * d] + (a.core[2 * d] > 0 ? 1 : 0);
    s1[d] = a.core[2 * d + 1] - (a.core[2 * d + 1] < n[d] ? 1 : 0);
    if (d == 0 && a.open_lo) s0[d] = std::max(s0[d], 1);          // the planes at an open
    if (d == 0 && a.open_hi) s1[d] = std::min(s1[d], n[d] - 1);   // face always have a pass 1
    s0[d] = std::min(std::max(s0[d], 0), n[d]);
    s1[d] = std::min(std::max(s1[d], s0[d]), n[d]);
  }
  boxes = {
      {0, s0[0], 0, n[1], 0, n[2]},           {s1[0], n[0], 0, n[1], 0, n[2]},
      {s0[0], s1[0], 0, s0[1], 0, n[2]},      {s0[0], s1[0], s1[1], n[1], 0, n[2]},
      {s0[0], s1[0], s0[1], s1[1], 0, s0[2]}, {s0[0], s1[0], s0[1], s1[1], s1[2], n[2]}};
  for (int m = 0; m < 6; ++m) hole[m] = 0;
  if (a.nobj > 0) {
    const CellBox h{std::max(a.hull[0] - 1, s0[0]), std::min(a.hull[1] + 1, s1[0]),
                    std::max(a.hull[2] - 1, s0[1]), std::min(a.hull[3] + 1, s1[1]),
                    std::max(a.hull[4] - 1, s0[2]), std::min(a.hull[5] + 1, s1[2])};
    boxes.push_back(h);
    if (h.i1 > h.i0 && h.j1 > h.j0 && h.k1 > h.k0) {
      hole[0] = h.i0; hole[1] = h.i1; hole[2] = h.j0; hole[3] = h.j1; hole[4] = h.k0; hole[5] = h.k1;
    }
  }
}

template <typename T>
int adjoint_shell_boxes(const fdtdhip_plan* p, const AdjArgs<T>& a, CellBoxes& tab, int (&s0)[3],
                        int (&s1)[3], int (&hole)[6]) {
  std::vector<CellBox> boxes;
  adjoint_shell_box_list<T>(p, a, boxes, s0, s1, hole);
  return cell_box_table(boxes, tab);
}

// Reverse of post(): sources (hard assignments kill the adjoint), then the boundaries that
// follow the first periodic one, last first.  stage < 0: all of it.  With a periodic boundary
// whose copy crosses ranks (the caller's) the work comes in two stages, as in the forward
// pass, in reverse order: stage 1 = the sources and the boundaries registered after that
// boundary, stage 0 = the boundaries registered before it.
template <typename T>
int adjoint_post(fdtdhip_plan* p, bool phaseE, T* lamE, T* lamH, int stage) {
  if (!p->inv_eps || !p->inv_mu) return fail(-1, "fields not set");
  if (!lamE || !lamH) return fail(-1, "null adjoint array");
  T* lamF = phaseE ? lamE : lamH;
  const T* mat = (const T*)(phaseE ? p->inv_eps : p->inv_mu);
  const size_t ext = external_periodic(p);
  if (stage >= 0 && ext == p->boundaries.size()) return fail(-1, "no external periodic boundary");
  if (stage < 0 && ext != p->boundaries.size())
    return fail(-1, "a periodic boundary that crosses ranks: drive fdtdhip_adjoint_post_stage around the caller's plane transfer");
  for (auto it = p->sources.rbegin(); it != p->sources.rend(); ++it) {
    if (stage == 0) break;
    const SourceH& s = *it;
    if (s.kind == FDTDHIP_SRC_SOFT_POINT) {
      // the voltage added to E is a constant plus impedance * (loop current of the step
      // before): its adjoint goes to the adjoint of H as the H half step before left it
      if (!phaseE || s.nwave < 0 || !(s.impedance > 0)) continue;
      // (on an x-slab with the cell on plane 0 the loop reaches into another rank's plane)
      if (p->H_below && s.elem0 < (int64_t)p->ny * p->nz * 3 && !s.adj_below)
        return fail(-1, "soft source on plane 0 of an x-slab: fdtdhip_plan_set_soft_source_adjoint_below "
                        "names the plane that collects the terms of the rank below");
      if (s.adj_q0 < 0)
        return fail(-1, "soft source: fdtdhip_plan_set_soft_source_adjoint names the time step of the adjoint's row 0");
      if (s.adj_q0 + p->trace_cursor <= 0) continue;   // no current yet at time step 0
      hipLaunchKernelGGL((adj_source_soft_point<T>), dim3(1), dim3(1), 0, p->stream, (const T*)lamE,
                         lamH, s.elem0, p->nx, p->ny, p->nz, s.impedance, s.spacing, (T*)s.adj_below);
      continue;
    }
    if (s.kind == FDTDHIP_SRC_MODULATED_BOX && phaseE && s.modP) {
      const int64_t row = p->trace_cursor;
      const int64_t cells =
          (int64_t)(s.box[1] - s.box[0]) * (s.box[3] - s.box[2]) * (s.box[5] - s.box[4]);
      if (cells <= 0) continue;
      if (!s.etrace || !s.gP || !s.gosc || !s.gwb || row < 0 || row >= s.osc_rows)
        return fail(-1, "the adjoint of a modulated source needs its energy trace and gradient arrays");
      hipLaunchKernelGGL((adj_source_modulated<T>), dim3(s.channels), dim3(256), 0, p->stream,
                         (const T*)lamF, p->ny, p->nz, s.box[0], s.box[1], s.box[2], s.box[3],
                         s.box[4], s.box[5], s.compE, (const T*)s.modP,
                         (const T*)s.osc + row * s.channels, (const T*)s.modw,
                         (const T*)s.etrace + row * cells * 2, (T*)s.gP,
                         (T*)s.gosc + row * s.channels, (T*)s.gwb);
      continue;
    }
    if (s.kind != FDTDHIP_SRC_SET_BOX) continue;
    const int64_t total =
        (int64_t)(s.box[1] - s.box[0]) * (s.box[3] - s.box[2]) * (s.box[5] - s.box[4]);
    if (total <= 0) continue;
    hipLaunchKernelGGL((source_set_box<T>), dim3(blocks_for(total)), dim3(256), 0, p->stream, lamF,
                       p->ny, p->nz, s.box[0], s.box[1], s.box[2], s.box[3], s.box[4], s.box[5],
                       phaseE ? s.compE : s.compH, T(0));
  }
  size_t first_periodic = p->boundaries.size();
  for (size_t b = 0; b < p->boundaries.size(); ++b)
    if (p->boundaries[b].kind == 1 && first_periodic == p->boundaries.size()) first_periodic = b;
  for (size_t b = p->boundaries.size(); b-- > first_periodic;) {
    const BoundaryRef& br = p->boundaries[b];
    if (stage >= 0 && (b == ext || (stage == 1) != (b > ext))) continue;
    if (br.kind == 1) {
      const int axis = br.idx;
      if (axis == FDTDHIP_PERIODIC_X_EXTERNAL) continue;   // (stage < 0 was refused above)
      const int n = axis == 0 ? p->nx : (axis == 1 ? p->ny : p->nz);
      if (n < 2) continue;
      const int64_t total = p->cells() / n * 3;
      const int dst = phaseE ? 0 : n - 1, src = phaseE ? n - 1 : 0;
      hipLaunchKernelGGL((adj_periodic_copy<T>), dim3(blocks_for(total)), dim3(256), 0, p->stream,
                         lamF, p->nx, p->ny, p->nz, axis, dst, src);
    } else {
      const PmlH& h = p->pmls[br.idx];
      PmlK<T> k = make_pmlk<T>(p, h, phaseE);
      k.psi = k.psi_out = (T*)(phaseE ? h.lamE : h.lamH);
      if (!k.psi) return fail(-1, "PML adjoint arrays not set");
      const int64_t slab = (int64_t)(h.axis == 0 ? h.t : p->nx) * k.sy * k.sz;
      hipLaunchKernelGGL((adj_pml_apply_slab<T>), dim3(blocks_for(slab)), dim3(256), 0, p->stream,
                         lamF, mat, k, p->nx, p->ny, p->nz, (T)p->sc, phaseE ? T(1) : T(-1));
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

template <typename T>
int adjoint_cells(fdtdhip_plan* p, bool phaseE, T* lamE, T* lamH, T* work) {
  if (!p->inv_eps || !p->inv_mu) return fail(-1, "fields not set");
  if (!lamE || !lamH || !work) return fail(-1, "null adjoint array");
  for (const ObjH& o : p->objs)
    if (o.kind == FDTDHIP_OBJ_ANISOTROPIC_MODULATED && (!o.modw || !o.etrace || !o.trace))
      return fail(-1, "the adjoint of a modulated object needs its curl and energy traces");
  if (int rc = sync_obj_table<T>(p)) return rc;
  // (with a periodic boundary that crosses ranks the caller has driven the two stages of
  // the post pieces around its plane transfer: fdtdhip_adjoint_post_stage)
  if (external_periodic(p) == p->boundaries.size())
    if (int rc = adjoint_post<T>(p, phaseE, lamE, lamH, -1)) return rc;
  AdjArgs<T> a;
  if (int rc = adjoint_args<T>(p, phaseE, lamE, lamH, work, a)) return rc;
  const dim3 block(256);
  const CellBoxes whole{};   // (not read by the kernels that cover the whole grid)
  if (a.direct) {
    CellBoxes tab;
    int s0[3], s1[3], hole[6];
    if (int rc = adjoint_shell_boxes<T>(p, a, tab, s0, s1, hole)) return rc;
    if (tab.count) {
      const dim3 grid((unsigned)tab.first[tab.count]);
      if (phaseE) hipLaunchKernelGGL((adj_E_cells<T, true, true>), grid, block, 0, p->stream, a, tab);
      else hipLaunchKernelGGL((adj_H_cells<T, true, true>), grid, block, 0, p->stream, a, tab);
    }
  } else if (adjoint_rows(p)) {
    const dim3 grid((p->nz + 63) / 64, (p->ny + 3) / 4, p->nx);
    if (phaseE) hipLaunchKernelGGL((adj_E_cells<T, true>), grid, block, 0, p->stream, a, whole);
    else hipLaunchKernelGGL((adj_H_cells<T, true>), grid, block, 0, p->stream, a, whole);
  } else {
    const dim3 grid(blocks_for(p->cells()));
    if (phaseE) hipLaunchKernelGGL((adj_E_cells<T, false>), grid, block, 0, p->stream, a, whole);
    else hipLaunchKernelGGL((adj_H_cells<T, false>), grid, block, 0, p->stream, a, whole);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

template <typename T>
int adjoint_gather(fdtdhip_plan* p, bool phaseE, T* lamE, T* lamH, T* work, const T* wghost) {
  if (!lamE || !lamH || !work) return fail(-1, "null adjoint array");
  AdjArgs<T> a;
  if (int rc = adjoint_args<T>(p, phaseE, lamE, lamH, work, a)) return rc;
  a.wghost = wghost;
  const dim3 block(256);
  if (adjoint_rows(p)) {
    const dim3 grid((p->nz + 63) / 64, (p->ny + 3) / 4, p->nx);
    if (phaseE) hipLaunchKernelGGL((adj_gather<T, true, true>), grid, block, 0, p->stream, a);
    else hipLaunchKernelGGL((adj_gather<T, false, true>), grid, block, 0, p->stream, a);
  } else {
    const dim3 grid(blocks_for(p->cells()));
    if (phaseE) hipLaunchKernelGGL((adj_gather<T, true, false>), grid, block, 0, p->stream, a);
    else hipLaunchKernelGGL((adj_gather<T, false, false>), grid, block, 0, p->stream, a);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- whole reverse steps (yee_adjoint_march.hpp) ---------------------------------------
// FDTDHIP_ADJ_MARCH: 0 = never, 2 = wherever the plan allows, whatever the grid size (tests);
// default: large 3-D grids.
int adjoint_march_mode() {
  static const int mode = [] {
    const char* e = getenv("FDTDHIP_ADJ_MARCH");
    return e ? atoi(e) : 1;
  }();
  return mode;
}

#ifndef FDTDHIP_ADJ_MARCH_TY32
#define FDTDHIP_ADJ_MARCH_TY32 4   // rows per workgroup of the reverse marching kernel in float32
#endif
#ifndef FDTDHIP_ADJ_MARCH_TY64
#define FDTDHIP_ADJ_MARCH_TY64 4   // the same in float64 (z rows of up to 512 cells)
#endif

// Whether the reverse sweep of this plan can take whole steps: second adjoint buffers, the
// plain cells in their direct form (constant materials), nothing that acts on the field
// adjoints between the half steps (no recorded periodic boundary and hence no PML whose
// correction follows one; only additive point sources, which have no adjoint of their own),
// a box of plain cells to march over.
// `staged`: asked for fdtdhip_adjoint_step_stage, the whole reverse step in pieces around the
// plane exchanges of an x-slab (open faces allowed).
template <typename T>
bool adjoint_march_eligible(fdtdhip_plan* p, const AdjArgs<T>& a, const int (&s0)[3], const int (&s1)[3],
                            bool staged = false) {
  const int mode = adjoint_march_mode();
  if (mode == 0 || !p->adjE_alt || !p->adjH_alt || !a.direct || p->no_alloc) return false;
  for (const PmlH& h : p->pmls)
    if (!h.lamE_alt || !h.lamH_alt) return false;
  if ((p->H_below || p->E_above) && !staged) return false;
  for (const BoundaryRef& b : p->boundaries)
    if (b.kind == 1) return false;
  for (const SourceH& s : p->sources)
    if (s.kind != FDTDHIP_SRC_ADD_POINTS) return false;
  if (p->nz > 64 * ADJ_MARCH_MAX_WAVES || p->nx / 16 > 65535) return false;
  for (int d = 0; d < 3; ++d)
    if (s1[d] - s0[d] < 1) return false;
  if (mode > 1) return true;
  const int64_t core = (int64_t)(s1[0] - s0[0]) * (s1[1] - s0[1]) * (s1[2] - s0[2]);
  return p->nz >= 128 && p->ny >= 64 && p->nx >= 64 && p->cells() >= (1ll << 22) &&
         core * 2 >= p->cells();
}

template <typename T, int TY, int MW>
int launch_adjoint_march_tiles(fdtdhip_plan* p, const AdjArgs<T>& e, const AdjArgs<T>& h, const T* lamE0,
                               const T* lamH0, T* lamE1, T* lamH1, const int (&s0)[3],
                               const int (&s1)[3], const int (&hole)[6], int part) {
  AdjMarchArgs<T> m;
  m.lamE0 = lamE0; m.lamH0 = lamH0; m.lamE1 = lamE1; m.lamH1 = lamH1;
  m.nx = p->nx; m.ny = p->ny; m.nz = p->nz;
  m.sc = (T)p->sc;
  for (int c = 0; c < 3; ++c) { m.bgE[c] = e.bg[c]; m.bgH[c] = h.bg[c]; }
  m.open_lo = e.open_lo; m.open_hi = e.open_hi;
  for (int d = 0; d < 3; ++d) { m.box[2 * d] = s0[d]; m.box[2 * d + 1] = s1[d]; }
  for (int n = 0; n < 6; ++n) m.hole[n] = hole[n];
  const bool holed = hole[1] > hole[0];
  const int tiles = (s1[1] - s0[1] + TY - 1) / TY, planes = s1[0] - s0[0];
  const int waves = (p->nz + 63) / 64;
  void (*kernel)(AdjMarchArgs<T>) = holed ? adj_march_box<T, TY, true, MW> : adj_march_box<T, TY, false, MW>;
  // part: 0 all chunks of planes in one launch, 1 / 2 the first / second half of them (each
  // half should fill its rounds of workgroups)
  const int lx = m.lx = march_chunk_planes(part == 0 ? planes : (planes + 1) / 2, tiles,
                                           march_slots(p, (const void*)kernel, 64 * waves));
  const int nchunks = (planes + lx - 1) / lx, half = nchunks / 2;
  m.band = tiles >= 16 ? (tiles + 7) / 8 : 0;
  m.chunk0 = part == 2 ? half : 0;
  const int count = part == 0 ? nchunks : (part == 1 ? half : nchunks - half);
  if (count > 0) {
    const dim3 grid(m.band ? 8 * m.band : tiles, count, 1), block(64 * waves);
    hipLaunchKernelGGL(kernel, grid, block, 0, p->stream, m);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

// The PML faces of a plan as arguments of adj_shell_faces, if the plan qualifies: at most one
// face per side of an axis, the two faces of an axis apart, and no object next to a slab (so
// that every cell those launches evaluate has the constant material).
template <typename T>
bool adjoint_face_set(const fdtdhip_plan* p, bool phaseE, int flip, const AdjArgs<T>& e, const int (&s0)[3],
                      const int (&s1)[3], AdjFaceSet<T>& fs) {
  const int n[3] = {p->nx, p->ny, p->nz};
  for (int ax = 0; ax < 3; ++ax)
    for (int side = 0; side < 2; ++side) fs.f[ax][side] = AdjFaceK<T>{nullptr, nullptr, nullptr, nullptr, 0, 0, 1, 1};
  for (const PmlH& h : p->pmls) {
    AdjFaceK<T>& f = fs.f[h.axis][h.lo == 0 ? 0 : 1];
    if (f.t != 0 || !h.fused) return false;
    T* first = (T*)(phaseE ? h.lamE : h.lamH);
    T* second = (T*)(phaseE ? h.lamE_alt : h.lamH_alt);
    f.psi = flip == 1 ? first : second;
    f.psi_out = flip == 1 ? second : first;
    f.b = (const T*)(phaseE ? h.bE : h.bH);
    f.c = (const T*)(phaseE ? h.cE : h.cH);
    f.lo = h.lo;
    f.t = h.t;
    f.sy = h.axis == 1 ? h.t : p->ny;
    f.sz = h.axis == 2 ? h.t : p->nz;
  }
  for (int ax = 0; ax < 3; ++ax)
    if (fs.f[ax][0].t + fs.f[ax][1].t > n[ax] || (fs.f[ax][0].t && fs.f[ax][0].lo != 0)) return false;
  if (e.nobj > 0)
    for (int d = 0; d < 3; ++d)
      if (e.hull[2 * d] <= s0[d] || e.hull[2 * d + 1] >= s1[d]) return false;
  return true;
}

// One whole reverse step, out of place: (lamE0, lamH0) after the step -> (lamE1, lamH1)
// before it; memory-variable adjoints from one set of arrays to the other (`flip`: 1 first ->
// second, 2 back).  In pieces (`stages`, a mask of AST_*), so that an x-slab can place them
// around its two plane exchanges -- the mirror image of the forward step in stages:
//   AST_H_FREE   H half step undone in the cells next to PML slabs, objects and the high face
//                of the slab (all but plane 0 of a slab that continues below)
//   AST_MARCH_A  adj_march_box over the first half of the planes of the plain cells
//   AST_H_GHOST  H half step undone in plane 0, which gathers from the plane of the rank below
//   AST_MARCH_B  the second half of the planes
//   AST_E_FREE   E half step undone in the cells next to PML slabs, objects and the low face
//                (all but the last plane of a slab that continues above)
//   AST_E_GHOST  E half step undone in the last plane, which gathers from the rank above
// The marching kernel reads the new lamE of plane 0 nowhere (its lowest plane is plane 1,
// which gathers from plane 2), so every operand is the one the two-pass sweep would use.
// Where the plan qualifies (adjoint_face_set) the slabs take adj_shell_faces and the box around
// the objects the two passes through `work` (adj_E_cells, adj_shell_gather);
// otherwise all of them take the table-walking adj_shell_H / adj_shell_E and
// adj_rescale_absorbing rescales lamE1 under absorbing objects at the end.
enum { AST_H_FREE = 1, AST_MARCH_A = 2, AST_H_GHOST = 4, AST_MARCH_B = 8, AST_E_FREE = 16, AST_E_GHOST = 32,
       AST_ALL = 63 };

template <typename T>
int adjoint_march_pieces(fdtdhip_plan* p, T* lamE0, T* lamH0, T* lamE1, T* lamH1, T* work, int flip,
                         int stages, const T* ghost) {
  AdjArgs<T> h, e;
  std::vector<CellBox> boxes;
  int s0[3], s1[3], hole[6];
  // (the boxes of the E half step: objects act in that half step only, but every cell the
  // marching kernel leaves out needs both half steps undone here)
  if (int rc = adjoint_args<T>(p, true, lamE1, lamH0, work, e, flip)) return rc;
  adjoint_shell_box_list<T>(p, e, boxes, s0, s1, hole);
  if (int rc = adjoint_args<T>(p, false, lamE0, lamH0, work, h, flip)) return rc;
  e.wghost = h.wghost = ghost;
  static const bool faces_off = [] {
    const char* v = getenv("FDTDHIP_ADJ_FACES");
    return v && atoi(v) == 0;
  }();
  AdjFaceSet<T> fe, fh;
  const bool faces = !faces_off && adjoint_face_set<T>(p, true, flip, e, s0, s1, fe) &&
                     adjoint_face_set<T>(p, false, flip, e, s0, s1, fh);
  const dim3 block(256);
  bool absorbing = false;
  for (const ObjH& o : p->objs) absorbing |= o.kind == FDTDHIP_OBJ_ABSORBING;
  const int64_t hull = (int64_t)std::max(e.hull[1] - e.hull[0], 0) * std::max(e.hull[3] - e.hull[2], 0) *
                       std::max(e.hull[5] - e.hull[4], 0);
  // one half step undone over the boxes `which` (indices into `boxes`)
  auto shell = [&](bool phaseE, std::initializer_list<int> which) -> int {
    const AdjArgs<T>& a = phaseE ? e : h;
    const T* in = phaseE ? lamH0 : lamE0;
    T* out = phaseE ? lamH1 : lamE1;
    std::vector<CellBox> sl, ob;
    for (int n : which)
      if (n < (int)boxes.size()) (faces && n == 6 ? ob : sl).push_back(boxes[n]);
    CellBoxes slabs, rest;
    if (int rc = cell_box_table(sl, slabs)) return rc;
    if (int rc = cell_box_table(ob, rest)) return rc;
    if (slabs.count) {
      const dim3 grid((unsigned)slabs.first[slabs.count]);
      if (faces) {
        if (phaseE) hipLaunchKernelGGL((adj_shell_faces<T, true>), grid, block, 0, p->stream, a, fe, in, out, slabs);
        else hipLaunchKernelGGL((adj_shell_faces<T, false>), grid, block, 0, p->stream, a, fh, in, out, slabs);
      } else {
        if (phaseE) hipLaunchKernelGGL((adj_shell_E<T>), grid, block, 0, p->stream, a, in, out, slabs);
        else hipLaunchKernelGGL((adj_shell_H<T>), grid, block, 0, p->stream, a, in, out, slabs);
      }
    }
    if (rest.count) {
      // the box around the objects alone (no PML slab near it): two passes in the E half step,
      // none but the gather in the H half step, where all of its cells are plain.  (On a
      // second stream beside the slab launches both slowed down: 0.36 against 0.33 ms per
      // reverse step at 256^3 with a 64^3 box.)
      const dim3 grid((unsigned)rest.first[rest.count]);
      if (phaseE) {
        hipLaunchKernelGGL((adj_E_cells<T, true, true>), grid, block, 0, p->stream, a, rest);
        hipLaunchKernelGGL((adj_shell_gather<T, true>), grid, block, 0, p->stream, a, in, out, rest);
      } else {
        hipLaunchKernelGGL((adj_shell_gather<T, false>), grid, block, 0, p->stream, a, in, out, rest);
      }
    }
    HIPCHK(hipGetLastError());
    return 0;
  };
  auto march = [&](int part) -> int {
    if ((p->nz + 63) / 64 <= 8)
      return launch_adjoint_march_tiles<T, sizeof(T) == 4 ? FDTDHIP_ADJ_MARCH_TY32 : FDTDHIP_ADJ_MARCH_TY64, 8>(
          p, e, h, lamE0, lamH0, lamE1, lamH1, s0, s1, hole, part);
    return launch_adjoint_march_tiles<T, sizeof(T) == 4 ? FDTDHIP_ADJ_MARCH_TY32 : 2, ADJ_MARCH_MAX_WAVES>(
        p, e, h, lamE0, lamH0, lamE1, lamH1, s0, s1, hole, part);
  };
  const bool lo = e.open_lo, hi = e.open_hi;   // plane 0 / the last plane wait for a neighbour's plane
  if (stages & AST_H_FREE) {
    if (lo) { if (int rc = shell(false, {1, 2, 3, 4, 5, 6})) return rc; }
    else if (int rc = shell(false, {0, 1, 2, 3, 4, 5, 6})) return rc;
  }
  const bool both = (stages & AST_MARCH_A) && (stages & AST_MARCH_B);
  if (both) { if (int rc = march(0)) return rc; }
  else if (stages & AST_MARCH_A) { if (int rc = march(1)) return rc; }
  if ((stages & AST_H_GHOST) && lo)
    if (int rc = shell(false, {0})) return rc;
  if (!both && (stages & AST_MARCH_B))
    if (int rc = march(2)) return rc;
  if (stages & AST_E_FREE) {
    if (hi) { if (int rc = shell(true, {0, 2, 3, 4, 5, 6})) return rc; }
    else if (int rc = shell(true, {0, 1, 2, 3, 4, 5, 6})) return rc;
  }
  if (stages & AST_E_GHOST) {
    if (hi)
      if (int rc = shell(true, {1})) return rc;
    // (lamE1 under absorbing objects: once every lane of adj_shell_E has read it; the two
    // passes of the other form rescale it in their pass 1)
    if (!faces && absorbing && hull > 0)
      hipLaunchKernelGGL((adj_rescale_absorbing<T>), dim3(blocks_for(hull)), block, 0, p->stream, e, lamE1);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

template <typename T>
int adjoint_march_step(fdtdhip_plan* p, T* lamE0, T* lamH0, T* lamE1, T* lamH1, T* work, int flip) {
  return adjoint_march_pieces<T>(p, lamE0, lamH0, lamE1, lamH1, work, flip, AST_ALL, (const T*)nullptr);
}

// Whether fdtdhip_adjoint_step_stage can walk this plan (second arrays set, see adjoint_march_eligible).
template <typename T>
int adjoint_stage_possible(fdtdhip_plan* p) {
  if (!p->inv_eps || !p->inv_mu) return 0;
  AdjArgs<T> a;
  std::vector<CellBox> boxes;
  int s0[3], s1[3], hole[6];
  if (sync_obj_table<T>(p)) return 0;
  for (const PmlH& h : p->pmls)
    if (!h.lamE || !h.lamH) return 0;
  if (adjoint_args<T>(p, true, (T*)nullptr, (T*)nullptr, (T*)nullptr, a)) return 0;
  adjoint_shell_box_list<T>(p, a, boxes, s0, s1, hole);
  return adjoint_march_eligible<T>(p, a, s0, s1, true) ? 1 : 0;
}

// One piece of a whole reverse step on an x-slab (fdtdhip_adjoint_step_stage).
template <typename T>
int adjoint_stage(fdtdhip_plan* p, int stage, T* lamE0, T* lamH0, T* lamE1, T* lamH1, T* work, int flip,
                  const T* ghost) {
  if (!lamE0 || !lamH0 || !lamE1 || !lamH1 || !work) return fail(-1, "null adjoint array");
  if (flip != 1 && flip != 2) return fail(-1, "flip is 1 (first -> second arrays) or 2 (back)");
  if (int rc = sync_obj_table<T>(p)) return rc;
  if (stage == 0 && !adjoint_stage_possible<T>(p))
    return fail(-1, "this plan cannot take whole reverse steps in stages (fdtdhip_plan_adjoint_whole_steps)");
  static const int pieces[4] = {AST_H_FREE | AST_MARCH_A, AST_H_GHOST, AST_MARCH_B | AST_E_FREE, AST_E_GHOST};
  return adjoint_march_pieces<T>(p, lamE0, lamH0, lamE1, lamH1, work, flip, pieces[stage], ghost);
}

// The plane a rank sends across an open face in a whole reverse step (adj_pack_plane).
template <typename T>
int adjoint_pack(fdtdhip_plan* p, bool phaseE, T* lamE, T* lamH, T* out, int flip) {
  if (!lamE || !lamH || !out) return fail(-1, "null adjoint array");
  if (int rc = sync_obj_table<T>(p)) return rc;
  AdjArgs<T> a;
  if (int rc = adjoint_args<T>(p, phaseE, lamE, lamH, (T*)nullptr, a, flip)) return rc;
  const int64_t plane = (int64_t)p->ny * p->nz;
  const int i = phaseE ? 0 : p->nx - 1;
  if (phaseE) hipLaunchKernelGGL((adj_pack_plane<T, true>), dim3(blocks_for(plane)), dim3(256), 0, p->stream, a, i, out);
  else hipLaunchKernelGGL((adj_pack_plane<T, false>), dim3(blocks_for(plane)), dim3(256), 0, p->stream, a, i, out);
  HIPCHK(hipGetLastError());
  return 0;
}

// The paired adjoint launches serve plans in which nothing acts on the field adjoints
// between a gather and the next cell pass: no recorded periodic boundary (and hence no PML
// whose correction follows one), only additive point sources (no adjoint of their own).
bool adjoint_pairs_eligible(const fdtdhip_plan* p) {
  static const int mode = [] {
    const char* e = getenv("FDTDHIP_FUSED");
    return e ? (atoi(e) ? 1 : 0) : -1;
  }();
  if (mode == 0 || adjoint_direct_mode() > 1) return false;
  // (above 2^19 cells the gather derives the difference adjoints of plain cells itself and the
  // cell pass leaves them out -- adjoint_args; the paired launches know nothing of that, so
  // they stop there whatever the environment asks: with FDTDHIP_FUSED set they used to go on
  // and returned wrong gradients on grids of more cells)
  if (p->cells() > (int64_t)(1 << 19)) return false;
  if (adjoint_rows(p)) return false;   // (never true below 2^19 cells by default; keeps the flat mapping)
  for (const BoundaryRef& b : p->boundaries)
    if (b.kind == 1) return false;
  for (const PmlH& h : p->pmls)
    if (!h.fused) return false;
  for (const SourceH& s : p->sources)
    if (s.kind != FDTDHIP_SRC_ADD_POINTS) return false;
  return true;
}

template <typename T>
int adjoint_run(fdtdhip_plan* p, int64_t nsteps, void* lamE, void* lamH, void* work) {
  if (!p->inv_eps || !p->inv_mu) return fail(-1, "fields not set");
  if (p->H_below || p->E_above)
    return fail(-1, "x-slabs: drive fdtdhip_adjoint_cells / fdtdhip_adjoint_gather per half step");
  if (nsteps <= 0) return 0;
  if (adjoint_pairs_eligible(p)) {
    // small grids: the gather of one half step and the cell pass of the next in one
    // launch (adj_gather_then_cells), alternating between two work arrays
    const int64_t bytes = p->cells() * 6 * (int64_t)sizeof(T);
    if (p->adj_work_bytes < bytes) {
      if (p->no_alloc) return fail(-6, "internal: second work array would have to be allocated");
      void* w2 = nullptr;
      HIPCHK(hipMalloc(&w2, (size_t)bytes));
      p->owned.push_back(w2);
      p->adj_work = w2;
      p->adj_work_bytes = bytes;
    }
    if (int rc = sync_obj_table<T>(p)) return rc;
    T* wa = (T*)work;
    T* wb = (T*)p->adj_work;
    const dim3 grid(blocks_for(p->cells())), block(256);
    p->trace_cursor = nsteps - 1;
    if (int rc = adjoint_cells<T>(p, false, (T*)lamE, (T*)lamH, wa)) return rc;   // H of the last step
    for (int64_t s = nsteps - 1; s >= 0; --s) {
      AdjArgs<T> g, c;
      p->trace_cursor = s;
      // gather of the H half step (completes lamE) + cell pass of the E half step
      if (int rc = adjoint_args<T>(p, false, (T*)lamE, (T*)lamH, wa, g)) return rc;
      if (int rc = adjoint_args<T>(p, true, (T*)lamE, (T*)lamH, wb, c)) return rc;
      hipLaunchKernelGGL((adj_gather_then_cells<T, false>), grid, block, 0, p->stream, g, c);
      if (s == 0) break;
      // gather of the E half step (completes lamH) + cell pass of the H half step before it
      if (int rc = adjoint_args<T>(p, true, (T*)lamE, (T*)lamH, wb, g)) return rc;
      if (int rc = adjoint_args<T>(p, false, (T*)lamE, (T*)lamH, wa, c)) return rc;
      hipLaunchKernelGGL((adj_gather_then_cells<T, true>), grid, block, 0, p->stream, g, c);
    }
    HIPCHK(hipGetLastError());
    p->trace_cursor = 0;
    if (int rc = adjoint_gather<T>(p, true, (T*)lamE, (T*)lamH, wb, (const T*)nullptr)) return rc;
    p->trace_cursor = -1;
    return 0;
  }
  int64_t s = nsteps - 1;
  auto two_pass_step = [&]() -> int {
    p->trace_cursor = s;
    for (int phase = 0; phase < 2; ++phase) {  // undo the H half step, then the E half step
      if (int rc = adjoint_cells<T>(p, phase == 1, (T*)lamE, (T*)lamH, (T*)work)) return rc;
      if (int rc = adjoint_gather<T>(p, phase == 1, (T*)lamE, (T*)lamH, (T*)work, (const T*)nullptr))
        return rc;
    }
    return 0;
  };
  {
    // whole reverse steps in pairs, first -> second adjoint buffers and back, so that the
    // result ends in the caller's arrays; with an odd count the last recorded step is undone
    // in the two-pass form first
    AdjArgs<T> a;
    CellBoxes tab;
    int s0[3], s1[3], hole[6];
    p->trace_cursor = s;
    if (int rc = sync_obj_table<T>(p)) return rc;
    if (int rc = adjoint_args<T>(p, true, (T*)lamE, (T*)lamH, (T*)work, a)) return rc;
    if (int rc = adjoint_shell_boxes<T>(p, a, tab, s0, s1, hole)) return rc;
    if (adjoint_march_eligible<T>(p, a, s0, s1)) {
      if (nsteps % 2) {
        if (int rc = two_pass_step()) return rc;
        --s;
      }
      for (; s >= 1; s -= 2) {
        p->trace_cursor = s;
        if (int rc = adjoint_march_step<T>(p, (T*)lamE, (T*)lamH, (T*)p->adjE_alt, (T*)p->adjH_alt, (T*)work, 1))
          return rc;
        p->trace_cursor = s - 1;
        if (int rc = adjoint_march_step<T>(p, (T*)p->adjE_alt, (T*)p->adjH_alt, (T*)lamE, (T*)lamH, (T*)work, 2))
          return rc;
      }
      p->trace_cursor = -1;
      return 0;
    }
  }
  for (; s >= 0; --s)
    if (int rc = two_pass_step()) return rc;
  p->trace_cursor = -1;
  return 0;
}

// ---- reverse sweep without allocations (fdtdhip_adjoint_run_noalloc) -----------------
// Everything fdtdhip_adjoint_run keeps in device memory besides the caller's arrays -- the
// `work` array of difference adjoints, the second one of the paired launches, the PML and
// object descriptor tables -- laid out in one static device array per GPU:
//   [0, 4 KiB)            PML descriptor tables
//   [4 KiB, 16 KiB)       object descriptor table (NOALLOC_MAX_OBJECTS entries)
//   [16 KiB, ...)         work, then the second work array, cells * 6 elements each
constexpr size_t NOALLOC_PML_OFF = 0, NOALLOC_OBJ_OFF = 4096, NOALLOC_WORK_OFF = 16384;
constexpr size_t NOALLOC_MAX_OBJECTS = 128;
static_assert(sizeof(PmlK<double>) * MAX_PML * TAB_COUNT <= NOALLOC_OBJ_OFF - NOALLOC_PML_OFF, "PML tables");
static_assert(sizeof(ObjK<double>) * NOALLOC_MAX_OBJECTS <= NOALLOC_WORK_OFF - NOALLOC_OBJ_OFF, "object table");
static_assert(NOALLOC_WORK_OFF % 256 == 0, "work arrays start on a 256-byte boundary");

}  // namespace

}  // namespace fdtdhip

// In host code this name is only a handle for hipGetSymbolAddress: every GPU holds its own
// copy of the array at its own address (noalloc_scratch below).
static __device__ __align__(256) unsigned char scratch[FDTDHIP_NOALLOC_SCRATCH_BYTES];

namespace fdtdhip {

namespace {

constexpr int NOALLOC_MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};
// Recorded on the stream of the last sweep that used the device's scratch array, behind its
// last kernel: the next sweep, on whatever plan and stream, starts behind it.
hipEvent_t g_scratch_idle[NOALLOC_MAX_DEVICES] = {};

// Device address of `scratch` on `device`, which must be the current device, and the event
// that orders its users.  Looked up / created once per device.
int noalloc_scratch(int device, unsigned char** out, hipEvent_t* idle) {
  if (device < 0 || device >= NOALLOC_MAX_DEVICES) return fail(-1, "device ordinal out of range");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void* ptr = nullptr;
    HIPCHK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(-2, "hipGetSymbolAddress returned no address for the scratch array");
    hipEvent_t ev = nullptr;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    g_scratch_idle[device] = ev;
    g_scratch_ptr[device] = (unsigned char*)ptr;
  }
  *out = g_scratch_ptr[device];
  *idle = g_scratch_idle[device];
  return 0;
}

template <typename T>
int adjoint_run_noalloc(fdtdhip_plan* p, int64_t nsteps, void* lamE, void* lamH) {
  const size_t work_bytes = (size_t)p->cells() * 6 * sizeof(T);
  const size_t work_stride = (work_bytes + 255) / 256 * 256;   // the second array starts aligned
  if (p->cells() > (int64_t)(FDTDHIP_NOALLOC_SCRATCH_BYTES / (12 * sizeof(T))) ||
      NOALLOC_WORK_OFF + work_stride + work_bytes > FDTDHIP_NOALLOC_SCRATCH_BYTES)
    return fail(-7, "fdtdhip_adjoint_run_noalloc: grid of " + std::to_string(p->cells()) +
                        " cells needs " + std::to_string(NOALLOC_WORK_OFF + work_stride + work_bytes) +
                        " bytes of scratch, the static array holds " +
                        std::to_string((size_t)FDTDHIP_NOALLOC_SCRATCH_BYTES) +
                        "; use fdtdhip_adjoint_run with a work array of your own");
  if (p->objs.size() > NOALLOC_MAX_OBJECTS)
    return fail(-7, "fdtdhip_adjoint_run_noalloc: more than " + std::to_string(NOALLOC_MAX_OBJECTS) +
                        " objects; use fdtdhip_adjoint_run");
  unsigned char* base = nullptr;
  hipEvent_t idle = nullptr;
  if (int rc = noalloc_scratch(p->device, &base, &idle)) return rc;
  // a sweep enqueued earlier on another stream may still read the array
  HIPCHK(hipStreamWaitEvent(p->stream, idle, 0));
  // the scratch array keeps whatever the last call left: both work arrays start at zero,
  // as the arrays callers hand to fdtdhip_adjoint_run do
  unsigned char* work = base + NOALLOC_WORK_OFF;
  HIPCHK(hipMemsetAsync(work, 0, work_stride + work_bytes, p->stream));
  // point the plan's tables at the scratch array for the duration of the sweep
  void* const obj_table = p->obj_table;
  const size_t obj_table_bytes = p->obj_table_bytes;
  const bool obj_dirty = p->obj_dirty, pml_dirty = p->pml_dirty;
  const int ovr_obj = p->ovr_obj;
  void* const pml_tables = p->pml_tables;
  void* const adj_work = p->adj_work;
  const int64_t adj_work_bytes = p->adj_work_bytes;
  p->obj_table = base + NOALLOC_OBJ_OFF;
  p->obj_table_bytes = sizeof(ObjK<T>) * p->objs.size();
  p->pml_tables = base + NOALLOC_PML_OFF;
  p->obj_dirty = p->pml_dirty = true;   // (re)written there by the first kernel set-up
  p->adj_work = work + work_stride;
  p->adj_work_bytes = (int64_t)work_bytes;
  p->no_alloc = true;
  const int rc = adjoint_run<T>(p, nsteps, lamE, lamH, work);
  p->no_alloc = false;
  p->obj_table = obj_table;
  p->obj_table_bytes = obj_table_bytes;
  p->pml_tables = pml_tables;
  p->obj_dirty = obj_dirty;             // the plan's own tables are as current as they were
  p->ovr_obj = ovr_obj;
  p->pml_dirty = pml_dirty;
  p->adj_work = adj_work;
  p->adj_work_bytes = adj_work_bytes;
  // (recorded after a failed sweep too: what it did enqueue still uses the array)
  HIPCHK(hipEventRecord(idle, p->stream));
  return rc;
}

}  // namespace

int adj_run(fdtdhip_plan* p, int64_t nsteps, void* lamE, void* lamH, void* work) {
  return DISPATCH(p, adjoint_run<float>(p, nsteps, lamE, lamH, work),
                  adjoint_run<double>(p, nsteps, lamE, lamH, work));
}

int adj_run_noalloc(fdtdhip_plan* p, int64_t nsteps, void* lamE, void* lamH) {
  return DISPATCH(p, adjoint_run_noalloc<float>(p, nsteps, lamE, lamH),
                  adjoint_run_noalloc<double>(p, nsteps, lamE, lamH));
}

int adj_cells(fdtdhip_plan* p, bool phaseE, void* lamE, void* lamH, void* work) {
  return DISPATCH(p, adjoint_cells<float>(p, phaseE, (float*)lamE, (float*)lamH, (float*)work),
                  adjoint_cells<double>(p, phaseE, (double*)lamE, (double*)lamH, (double*)work));
}

int adj_gather(fdtdhip_plan* p, bool phaseE, void* lamE, void* lamH, void* work, const void* wghost) {
  return DISPATCH(p, adjoint_gather<float>(p, phaseE, (float*)lamE, (float*)lamH, (float*)work, (const float*)wghost),
                  adjoint_gather<double>(p, phaseE, (double*)lamE, (double*)lamH, (double*)work, (const double*)wghost));
}

int adj_stage_possible(fdtdhip_plan* p) {
  return DISPATCH(p, adjoint_stage_possible<float>(p), adjoint_stage_possible<double>(p));
}

int adj_stage(fdtdhip_plan* p, int stage, void* lamE0, void* lamH0, void* lamE1, void* lamH1, void* work,
              int flip, const void* ghost) {
  return DISPATCH(p, adjoint_stage<float>(p, stage, (float*)lamE0, (float*)lamH0, (float*)lamE1, (float*)lamH1,
                                          (float*)work, flip, (const float*)ghost),
                  adjoint_stage<double>(p, stage, (double*)lamE0, (double*)lamH0, (double*)lamE1, (double*)lamH1,
                                        (double*)work, flip, (const double*)ghost));
}

int adj_pack(fdtdhip_plan* p, bool phaseE, void* lamE, void* lamH, void* out, int flip) {
  return DISPATCH(p, adjoint_pack<float>(p, phaseE, (float*)lamE, (float*)lamH, (float*)out, flip),
                  adjoint_pack<double>(p, phaseE, (double*)lamE, (double*)lamH, (double*)out, flip));
}

int adj_post(fdtdhip_plan* p, bool phaseE, void* lamE, void* lamH, int stage) {
  return DISPATCH(p, adjoint_post<float>(p, phaseE, (float*)lamE, (float*)lamH, stage),
                  adjoint_post<double>(p, phaseE, (double*)lamE, (double*)lamH, stage));
}

}  // namespace fdtdhip
