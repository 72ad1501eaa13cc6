// Reverse sweep of libfdtd_hip: the adjoint of the half steps (yee_adjoint.hpp) driven
// backwards through time, stands in for torch autograd over the reference's step loop
// (reference examples/learnable_focus.py:98-128).
#include "plan_host.hpp"
#include "yee_generic.hpp"
#include "yee_tiled.hpp"
#include "yee_adjoint.hpp"

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

template <typename T>
int adjoint_args(fdtdhip_plan* p, bool phaseE, T* lamE, T* lamH, T* work, AdjArgs<T>& a) {
  a.lamF = phaseE ? lamE : lamH;
  a.lamG = phaseE ? lamH : lamE;
  a.mat = (const T*)(phaseE ? p->inv_eps : p->inv_mu);
  a.work = work;
  a.nx = p->nx; a.ny = p->ny; a.nz = p->nz;
  a.sc = (T)p->sc;
  a.npml = (int)p->pmls.size();
  a.nobj = phaseE ? (int)p->objs.size() : 0;
  a.obj = (const ObjK<T>*)p->obj_table;
  a.trace_step = phaseE ? p->trace_cursor : -1;
  for (const PmlH& h : p->pmls)
    if (!(phaseE ? h.lamE : h.lamH)) return fail(-1, "PML adjoint arrays not set");
  if (int rc = sync_pml_tables<T>(p)) return rc;
  a.pml = pml_table<T>(p, phaseE ? TAB_ADJ_E : TAB_ADJ_H);
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
      if (p->H_below || p->E_above)
        return fail(-1, "the adjoint does not cover a source fed back from a current detector on x-slabs");
      if (s.adj_q0 < 0)
        return fail(-1, "soft source: fdtdhip_plan_set_soft_source_adjoint names the time step of the adjoint's row 0");
      if (s.adj_q0 + p->trace_cursor <= 0) continue;   // no current yet at time step 0
      hipLaunchKernelGGL((adj_source_soft_point<T>), dim3(1), dim3(1), 0, p->stream, (const T*)lamE,
                         lamH, s.elem0, p->nx, p->ny, p->nz, s.impedance, s.spacing);
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
    // Only cells near a PML slab or an object have a pass 1 (the kernel leaves the others
    // out cell by cell): launch it over boxes that cover them instead of the whole grid --
    // six slabs around the part of the PML-free core whose 3x3x3 blocks stay inside it,
    // and the hull of the objects grown by one cell where it lies in that part.  The boxes
    // are disjoint: pass 1 has side effects (memory-variable adjoints, object gradients).
    int s0[3], s1[3];
    const int n[3] = {p->nx, p->ny, p->nz};
    for (int d = 0; d < 3; ++d) {
      s0[d] = a.core[2 * d] + (a.core[2 * d] > 0 ? 1 : 0);
      s1[d] = a.core[2 * d + 1] - (a.core[2 * d + 1] < n[d] ? 1 : 0);
      if (d == 0 && a.open_lo) s0[d] = std::max(s0[d], 1);          // the planes at an open
      if (d == 0 && a.open_hi) s1[d] = std::min(s1[d], n[d] - 1);   // face always have a pass 1
      s0[d] = std::min(std::max(s0[d], 0), n[d]);
      s1[d] = std::min(std::max(s1[d], s0[d]), n[d]);
    }
    std::vector<CellBox> boxes = {
        {0, s0[0], 0, n[1], 0, n[2]},           {s1[0], n[0], 0, n[1], 0, n[2]},
        {s0[0], s1[0], 0, s0[1], 0, n[2]},      {s0[0], s1[0], s1[1], n[1], 0, n[2]},
        {s0[0], s1[0], s0[1], s1[1], 0, s0[2]}, {s0[0], s1[0], s0[1], s1[1], s1[2], n[2]}};
    if (a.nobj > 0)
      boxes.push_back({std::max(a.hull[0] - 1, s0[0]), std::min(a.hull[1] + 1, s1[0]),
                       std::max(a.hull[2] - 1, s0[1]), std::min(a.hull[3] + 1, s1[1]),
                       std::max(a.hull[4] - 1, s0[2]), std::min(a.hull[5] + 1, s1[2])});
    CellBoxes tab;
    tab.count = 0;
    tab.first[0] = 0;
    for (const CellBox& b : boxes) {
      if (b.i1 <= b.i0 || b.j1 <= b.j0 || b.k1 <= b.k0) continue;
      int gx, gy;
      box_grid(b, gx, gy);
      tab.box[tab.count] = b;
      tab.gx[tab.count] = gx;
      tab.gy[tab.count] = gy;
      tab.first[tab.count + 1] = tab.first[tab.count] + (int64_t)gx * gy * (b.i1 - b.i0);
      ++tab.count;
    }
    for (int n = tab.count; n < MAX_CELL_BOXES; ++n) {
      tab.box[n] = CellBox{0, 0, 0, 0, 0, 0};
      tab.gx[n] = tab.gy[n] = 1;
      tab.first[n + 1] = tab.first[tab.count];
    }
    if (tab.first[tab.count] >= (1ll << 31)) return fail(-1, "internal: pass 1 of the adjoint over too many workgroups");
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

// The paired adjoint launches serve plans in which nothing acts on the field adjoints
// between a gather and the next cell pass: no recorded periodic boundary (and hence no PML
// whose correction follows one), only additive point sources (no adjoint of their own).
bool adjoint_pairs_eligible(const fdtdhip_plan* p) {
  static const int mode = [] {
    const char* e = getenv("FDTDHIP_FUSED");
    return e ? (atoi(e) ? 1 : 0) : -1;
  }();
  if (mode == 0 || adjoint_direct_mode() > 1) return false;
  if (mode < 0 && p->cells() > (int64_t)(1 << 19)) return false;
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
  for (int64_t s = nsteps - 1; s >= 0; --s) {
    p->trace_cursor = s;
    for (int phase = 0; phase < 2; ++phase) {  // undo the H half step, then the E half step
      if (int rc = adjoint_cells<T>(p, phase == 1, (T*)lamE, (T*)lamH, (T*)work)) return rc;
      if (int rc = adjoint_gather<T>(p, phase == 1, (T*)lamE, (T*)lamH, (T*)work, (const T*)nullptr))
        return rc;
    }
  }
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

int adj_post(fdtdhip_plan* p, bool phaseE, void* lamE, void* lamH, int stage) {
  return DISPATCH(p, adjoint_post<float>(p, phaseE, (float*)lamE, (float*)lamH, stage),
                  adjoint_post<double>(p, phaseE, (double*)lamE, (double*)lamH, stage));
}

}  // namespace fdtdhip
