// Host side of a plan, shared by the translation units of libfdtd_hip: the plan structure
// with its boundary / object / source / detector lists, error reporting, the descriptor
// tables the kernels read, and the entry points each unit offers the others.
//
//   plan_host.hip       tables, staging, timing, waveforms (no kernels)
//   launch_half.hip     half steps: bulk kernels and the post pieces (yee_tiled, yee_generic)
//   launch_whole.hip    whole steps: step_fused, the marching kernels and their shell
//   launch_adjoint.hip  reverse sweep
//   fdtd_hip.hip        the C ABI (include/fdtd_hip.h)
#pragma once

#include "../../include/fdtd_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "plan.hpp"

namespace fdtdhip {

// Records the message of a failure for fdtdhip_last_error (per thread) and returns `code`.
int fail(int code, const std::string& msg);
const char* last_error();

static const char* const kStaged =
    "a step taken in stages (fdtdhip_step_stage) is in progress: finish it with the remaining stages first";

static const char* const kPoisoned =
    "an earlier fused step failed half way: the plan's state is split between its two buffer "
    "sets; call fdtdhip_plan_reset (and set the fields again) before stepping";

#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return fail(-2, std::string(#expr) + ": " + hipGetErrorString(_e));                 \
  } while (0)

// Makes the plan's device current for the duration of a call.  `ok` is false when that
// could not be done: nothing may be enqueued then (it would land on another device).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (prev != dev) {
      switched = hipSetDevice(dev) == hipSuccess;
      ok = switched;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

#define GUARD(p)                 \
  DeviceGuard g((p)->device);    \
  if (!g.ok) return fail(-2, "could not make the plan's device current (hipSetDevice)")

struct PmlH {
  int axis, side, t, lo;
  void* psiE;
  void* psiH;
  void *bE, *cE, *bH, *cH;  // device arrays of the plan's element type
  bool fused;               // no periodic boundary precedes it in the list
  void* psiE_alt = nullptr; // second psiE buffer of the fused full step
  void* lamE = nullptr;     // adjoints of psiE / psiH (differentiable runs)
  void* lamH = nullptr;
  void* lamE_alt = nullptr; // second adjoint arrays of whole reverse steps
  void* lamH_alt = nullptr;
};

struct BoundaryRef {
  int kind;  // 0 = pml (index into pmls), 1 = periodic (axis)
  int idx;
};

struct ObjH {
  int kind;
  int box[6];
  const void* inv;
  const void* absorb;
  void* trace = nullptr;       // (capacity, box, 3) curl trace of a differentiable run
  int64_t trace_capacity = 0;
  void* grad = nullptr;        // gradient accumulator for `inv`
  const void* modw = nullptr;  // kind 3: (9,3) modifier weights
  void* etrace = nullptr;      // kind 3: (capacity, box, 2) energy trace
  void* gmod = nullptr;        // kind 3: (box,9,3) gradient contributions to modw
};

struct SourceH {
  int kind, compE, compH;
  int64_t n;
  int box[6];
  int64_t* elem = nullptr;    // device: flat element index of (cell, compE)
  int64_t elem_min = 0, elem_max = -1;
  std::vector<int64_t> elem_host;     // the same lists on the host (fused full step)
  std::vector<double> profile_host;
  double* profile = nullptr;  // device
  double amplitude = 0;       // SET_BOX
  int serial = 0;
  double period, phase, frequency, hanning_dt, cycle;
  int pulse;
  // FDTDHIP_SRC_MODULATED_BOX (fdtdhip_plan_set_modulated_source)
  int channels = 0;
  const void *modP = nullptr, *osc = nullptr, *modw = nullptr;
  int64_t osc_rows = 0, q0 = 0;
  void *etrace = nullptr, *gP = nullptr, *gosc = nullptr, *gwb = nullptr;
  // FDTDHIP_SRC_SOFT_POINT (fdtdhip_plan_add_soft_source)
  double* wave = nullptr;      // device copy of the voltage waveform
  int64_t nwave = 0, elem0 = 0;
  double impedance = 0, spacing = 1;
  int det = -1;                // its loop-current detector (fdtdhip_plan_link_soft_source)
  const void* prev_current = nullptr;  // last reading taken before this run, or null
  char *vin = nullptr, *vout = nullptr;
  int64_t vcap = 0, vused = 0;
  int64_t adj_q0 = -1;         // time step of row 0 of the adjoint calls (fdtdhip_plan_set_soft_source_adjoint)
  void* adj_below = nullptr;   // x-slabs: (ny,nz,3) plane that collects the adjoint terms of plane x - 1
};

struct DetH {
  int kind;
  int64_t n;
  int64_t* cell = nullptr;  // device flat cell index (CELLS)
  std::vector<int64_t> cell_host;
  int32_t* xyz = nullptr;   // device (CURRENT)
  const void* below = nullptr;  // CURRENT on an x-slab: the H plane under plane 0; read on request only
  double spacing;
  char* outE = nullptr;
  char* outH = nullptr;
  int64_t capacity = 0, usedE = 0, usedH = 0;
};

}  // namespace fdtdhip

// Pinned host staging for the small descriptor tables: a table is copied to the device with a
// stream-ordered copy out of one of a few pinned buffers, so that the host does not have to
// wait for the stream (a differentiable training loop re-points traces and state arrays for
// every recorded step).  A buffer is reused only after the copy out of it has completed.
struct StagingRing {
  static constexpr int SLOTS = 8;
  void* host[SLOTS] = {};
  size_t cap[SLOTS] = {};
  hipEvent_t done[SLOTS] = {};
  int next = 0;
};

struct fdtdhip_plan {
  StagingRing staging;
  int nx, ny, nz, dtype, device;
  double sc;
  hipStream_t stream = nullptr;
  void *E = nullptr, *H = nullptr;
  const void *inv_eps = nullptr, *inv_mu = nullptr;
  const void *H_below = nullptr, *E_above = nullptr;
  void *E_alt = nullptr, *H_alt = nullptr;  // second field pair of the fused full step
  bool step_flipped = false;                // an odd number of fused steps has been taken
  bool no_alloc = false;                    // inside fdtdhip_adjoint_run_noalloc: device tables
                                            // live in the static scratch array, nothing may
                                            // be allocated or freed
  int stage_next = 0;                       // fdtdhip_step_stage: the piece of the step due next
                                            // (0: no step in progress)
  bool poisoned = false;                    // a fused step failed while the state lived in the
                                            // second buffers: only reset / destroy are accepted
  std::vector<fdtdhip::BoundaryRef> boundaries;
  std::vector<fdtdhip::PmlH> pmls;
  std::vector<fdtdhip::ObjH> objs;
  std::vector<fdtdhip::SourceH> sources;
  std::vector<fdtdhip::DetH> dets;
  void* obj_table = nullptr;  // device ObjK<T>[]
  size_t obj_table_bytes = 0;
  bool obj_dirty = false;
  // the one object whose trace / gradient arrays travel with the kernel arguments (BulkArgs::ovr_obj)
  int ovr_obj = -1;
  // device tables of PML descriptors, one per use (see PmlTab); rebuilt when dirty
  void* pml_tables = nullptr;
  bool pml_dirty = true;
  // Outcome of fdtdhip_plan_scan_materials for inv_eps [0] and inv_mu [1]: the array is
  // mat_bg outside the object boxes (zero inside, inv_eps only) and need not be read.
  bool mat_const[2] = {false, false};
  double mat_bg[2][3] = {{0, 0, 0}, {0, 0, 0}};
  int* scan_flag = nullptr;   // device word the scan kernel raises
  // cell -> (detector, point) map for the fused step's in-kernel read-out; 0: not built,
  // 1: built, -1: not possible (shared cells, too many detectors or points)
  int32_t* detmap = nullptr;
  int detmap_state = 0;
  void *adjE_alt = nullptr, *adjH_alt = nullptr;   // second adjoint fields of whole reverse steps
  void* adj_work = nullptr;   // second `work` array of the paired adjoint launches (owned)
  int64_t adj_work_bytes = 0;
  int64_t trace_cursor = -1;  // row of the curl traces the next E half step writes; -1: no traces
  std::vector<void*> owned;   // device allocations released by reset/destroy
  // optional kernel timing (fdtdhip_plan_enable_timing)
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  std::vector<int> ev_phase;  // what each used pair brackets (TIME_*)
  double ms[5] = {0, 0, 0, 0, 0};
  int64_t launches[5] = {0, 0, 0, 0, 0};
  size_t esize() const { return dtype == FDTDHIP_F32 ? 4 : 8; }
  int64_t cells() const { return (int64_t)nx * ny * nz; }
};

namespace fdtdhip {

inline unsigned blocks_for(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

#define DISPATCH(p, call_f32, call_f64) ((p)->dtype == FDTDHIP_F32 ? (call_f32) : (call_f64))

// ---- plan_host.hip ----------------------------------------------------------------
int release_tables(fdtdhip_plan* p);
int upload_table(fdtdhip_plan* p, void* dst, const void* src, size_t bytes);
int drain_timing(fdtdhip_plan* p);
int span_begin(fdtdhip_plan* p, size_t& slot);
int span_end(fdtdhip_plan* p, size_t slot, int what);
double waveform(const SourceH& s, int64_t q);
// Whole-step kernels add up to MAX_STEP_SOURCES source points in registers.  Plans with more
// points, or with a source that sets its box (FDTDHIP_SRC_SET_BOX), apply ALL their sources by
// launches of their own after the half steps, as fdtdhip_update_E / _H do, in the order of the
// plan; the box that holds the source cells then belongs to the cells the shell launches step
// (march_core).  False also where a source keeps a plan from whole steps altogether.
bool sources_by_launch(const fdtdhip_plan* p);
// Periodic boundaries whose two planes both lie on this grid: in whole steps their planes are
// stepped by the shell launches and the copy (reference fdtd/boundaries.py:173-208) is a launch
// of its own after each half step, before the sources, as in fdtdhip_update_E / _H.
bool periodic_by_launch(const fdtdhip_plan* p);
int apply_periodic_to(fdtdhip_plan* p, bool phaseE, void* F);
// The sources of one half step at time index q applied to the field array F (which need not
// be the plan's current one), in the order of the plan.
int apply_sources_to(fdtdhip_plan* p, bool phaseE, int64_t q, void* F);
void advance_trace_cursor(fdtdhip_plan* p);
int64_t trace_rows_left(const fdtdhip_plan* p);   // rows the curl traces can still take; huge when none is recorded
extern int64_t two_step_launches;   // launches of the two-step kernels in this process (diagnostics)
int capacity_check(const fdtdhip_plan* p, int64_t n);
size_t external_periodic(const fdtdhip_plan* p);

template <typename T>
int upload_profile(fdtdhip_plan* p, const double* host, int n, void** out) {
  std::vector<T> tmp(n);
  for (int i = 0; i < n; ++i) tmp[i] = (T)host[i];
  void* d = nullptr;
  HIPCHK(hipMalloc(&d, sizeof(T) * (size_t)n));
  p->owned.push_back(d);
  HIPCHK(hipMemcpy(d, tmp.data(), sizeof(T) * (size_t)n, hipMemcpyHostToDevice));
  *out = d;
  return 0;
}

template <typename T>
int sync_obj_table(fdtdhip_plan* p) {
  if (!p->obj_dirty) return 0;
  std::vector<ObjK<T>> tab(p->objs.size());
  for (size_t o = 0; o < p->objs.size(); ++o) {
    const ObjH& h = p->objs[o];
    tab[o].inv = (const T*)h.inv;
    tab[o].absorb = (const T*)h.absorb;
    tab[o].trace = (T*)h.trace;
    tab[o].grad = (T*)h.grad;
    tab[o].modw = (const T*)h.modw;
    tab[o].etrace = (T*)h.etrace;
    tab[o].gmod = (T*)h.gmod;
    tab[o].kind = h.kind;
    tab[o].x0 = h.box[0]; tab[o].x1 = h.box[1];
    tab[o].y0 = h.box[2]; tab[o].y1 = h.box[3];
    tab[o].z0 = h.box[4]; tab[o].z1 = h.box[5];
  }
  const size_t bytes = sizeof(ObjK<T>) * tab.size();
  if (p->no_alloc && !tab.empty() && (!p->obj_table || p->obj_table_bytes != bytes))
    return fail(-6, "internal: object table would have to be allocated");
  if (!p->no_alloc && p->obj_table && p->obj_table_bytes != bytes) {
    // the old table may still be read by kernels in flight on the plan's stream
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipFree(p->obj_table));
    p->obj_table = nullptr;
  }
  if (!tab.empty()) {
    if (!p->obj_table) {
      HIPCHK(hipMalloc(&p->obj_table, bytes));
      p->obj_table_bytes = bytes;
    }
    if (int rc = upload_table(p, p->obj_table, tab.data(), bytes)) return rc;
  }
  p->obj_dirty = false;
  p->ovr_obj = -1;      // (the table holds every array again)
  return 0;
}

// The object whose trace / gradient arrays go with the kernel arguments (plan.hpp: load_obj).
template <typename T, typename Args>
void fill_object_override(const fdtdhip_plan* p, Args& a) {
  const bool on = p->ovr_obj >= 0 && p->ovr_obj < (int)p->objs.size();
  a.ovr_obj = on ? p->ovr_obj : -1;
  a.ovr_trace = on ? (T*)p->objs[p->ovr_obj].trace : nullptr;
  a.ovr_grad = on ? (T*)p->objs[p->ovr_obj].grad : nullptr;
}

// Which PML descriptor table a kernel reads: the two half steps in place, the E half
// of the fused full step in its two directions (first -> second psi_E buffer and
// back), the adjoint half steps (psi -> adjoint arrays) in place, and from the first to the
// second adjoint arrays and back (whole reverse steps).
enum PmlTab { TAB_E = 0, TAB_H, TAB_E_AB, TAB_E_BA, TAB_ADJ_E, TAB_ADJ_H,
              TAB_ADJ_E_AB, TAB_ADJ_E_BA, TAB_ADJ_H_AB, TAB_ADJ_H_BA, TAB_COUNT };

template <typename T>
PmlK<T> make_pmlk(const fdtdhip_plan* p, const PmlH& h, bool phaseE);

template <typename T>
int sync_pml_tables(fdtdhip_plan* p) {
  if (!p->pml_dirty) return 0;
  const size_t bytes = sizeof(PmlK<T>) * MAX_PML * TAB_COUNT;
  if (!p->pml_tables) {
    if (p->no_alloc) return fail(-6, "internal: PML tables would have to be allocated");
    HIPCHK(hipMalloc(&p->pml_tables, bytes));
  }
  std::vector<PmlK<T>> host(MAX_PML * TAB_COUNT);
  for (size_t i = 0; i < p->pmls.size(); ++i) {
    const PmlH& h = p->pmls[i];
    PmlK<T> e = make_pmlk<T>(p, h, true), hh = make_pmlk<T>(p, h, false);
    host[TAB_E * MAX_PML + i] = e;
    host[TAB_H * MAX_PML + i] = hh;
    PmlK<T> ab = e, ba = e;
    ab.psi_out = (T*)h.psiE_alt;
    ba.psi = (T*)h.psiE_alt;
    ba.psi_out = (T*)h.psiE;
    host[TAB_E_AB * MAX_PML + i] = ab;
    host[TAB_E_BA * MAX_PML + i] = ba;
    PmlK<T> ae = e, ah = hh;
    ae.psi = ae.psi_out = (T*)h.lamE;
    ah.psi = ah.psi_out = (T*)h.lamH;
    host[TAB_ADJ_E * MAX_PML + i] = ae;
    host[TAB_ADJ_H * MAX_PML + i] = ah;
    PmlK<T> eab = ae, eba = ae, hab = ah, hba = ah;
    eab.psi_out = (T*)h.lamE_alt;
    eba.psi = (T*)h.lamE_alt;
    hab.psi_out = (T*)h.lamH_alt;
    hba.psi = (T*)h.lamH_alt;
    host[TAB_ADJ_E_AB * MAX_PML + i] = eab;
    host[TAB_ADJ_E_BA * MAX_PML + i] = eba;
    host[TAB_ADJ_H_AB * MAX_PML + i] = hab;
    host[TAB_ADJ_H_BA * MAX_PML + i] = hba;
  }
  if (int rc = upload_table(p, p->pml_tables, host.data(), bytes)) return rc;
  p->pml_dirty = false;
  return 0;
}

template <typename T>
const PmlK<T>* pml_table(const fdtdhip_plan* p, int which) {
  return (const PmlK<T>*)p->pml_tables + which * MAX_PML;
}

template <typename T>
PmlK<T> make_pmlk(const fdtdhip_plan* p, const PmlH& h, bool phaseE) {
  PmlK<T> k;
  k.psi = (T*)(phaseE ? h.psiE : h.psiH);
  k.psi_out = k.psi;
  k.b = (const T*)(phaseE ? h.bE : h.bH);
  k.c = (const T*)(phaseE ? h.cE : h.cH);
  k.axis = h.axis;
  k.lo = h.lo;
  k.t = h.t;
  k.fused = h.fused ? 1 : 0;
  k.sy = h.axis == 1 ? h.t : p->ny;
  k.sz = h.axis == 2 ? h.t : p->nz;
  return k;
}

// Kernel strategy, overridable for experiments with FDTDHIP_BULK, a '+'-separated
// list of: generic (branchy one-lane-per-cell kernel), band (its XCD-banded
// variant), flat | rows (lane mapping of the fast kernel), ntl / nts
// (non-temporal loads of the streamed arrays / non-temporal stores), nont.
// FDTDHIP_FORCE=1 applies the fast kernel to any grid size.
struct Strategy {
  bool chosen = false;  // FDTDHIP_BULK given: use it for every element type
  bool fast = true;
  bool flat = true;
  bool ntl = true, nts = true;
  bool band = false;
  bool force = false;
};

const Strategy& strategy();

// What a pair of timing events brackets: a bulk half-step launch (H, E), a whole-step kernel
// (step_march, step_march_box, step_fused), or the half-step launches over the shell that
// accompany the marching kernel on plans with PMLs, objects or open slab faces.
enum { TIME_H = 0, TIME_E = 1, TIME_WHOLE = 2, TIME_SHELL = 3, TIME_PAIR = 4, TIME_KINDS = 5 };

// The box of cells that no PML slab covers (slabs sit at the low or high end of their
// axis) and the box that contains all objects.
template <typename T>
void fill_core(const fdtdhip_plan* p, BulkArgs<T>& a) {
  int c[6] = {0, p->nx, 0, p->ny, 0, p->nz};
  for (const PmlH& h : p->pmls) {
    if (h.lo == 0) c[2 * h.axis] = std::max(c[2 * h.axis], h.t);
    else c[2 * h.axis + 1] = std::min(c[2 * h.axis + 1], h.lo);
  }
  for (int n = 0; n < 6; ++n) a.core[n] = c[n];
  for (int n = 0; n < MAX_PML; ++n) {
    const bool have = n < (int)p->pmls.size();
    a.face_axis[n] = have ? p->pmls[n].axis : 0;
    a.face_lo[n] = have ? p->pmls[n].lo : 0;
    a.face_hi[n] = have ? p->pmls[n].lo + p->pmls[n].t : 0;
  }
  int h[6] = {p->nx, 0, p->ny, 0, p->nz, 0};
  for (const ObjH& o : p->objs)
    for (int ax = 0; ax < 3; ++ax) {
      h[2 * ax] = std::min(h[2 * ax], o.box[2 * ax]);
      h[2 * ax + 1] = std::max(h[2 * ax + 1], o.box[2 * ax + 1]);
    }
  for (int n = 0; n < 6; ++n) a.hull[n] = h[n];
  // z segment of the row kernels rotated every 2^rot block rows (lane_cell).  MI355X, two
  // half-step launches per step with six PMLs, rot = 2 / 3 / 4: 512^3 float32 (8 segments)
  // 1.735 / 1.794 / 1.751 ms; 256^3 float32 (4 segments) 0.237 / 0.235 / 0.233; float64 256^3
  // 0.493 / 0.479 / 0.477 and 512^3 3.78 / 3.74 / 3.73 -- every 16 rows, except float32 rows of
  // more than four segments.
  static const int rot = [] {
    const char* e = getenv("FDTDHIP_ROT");
    return e ? atoi(e) : -1;
  }();
  a.rot_shift = rot >= 0 ? rot : (sizeof(T) == 8 || (p->nz + 63) / 64 <= 4 ? 4 : 2);
}

template <typename T>
void fill_material(const fdtdhip_plan* p, bool phaseE, BulkArgs<T>& a) {
  const int w = phaseE ? 0 : 1;
  a.mat_const = p->mat_const[w] ? 1 : 0;
  for (int c = 0; c < 3; ++c) a.bg[c] = (T)p->mat_bg[w][c];
}

// The pieces of a half step that follow the bulk pass, selectable so that a run can
// interleave plane chunks of the two half steps: boundaries (periodic copies, PMLs
// registered after a periodic boundary), sources restricted to planes [x0, x1),
// and the end-of-half-step work (detectors, curl-trace cursor).
enum { POST_BOUNDARIES = 1, POST_SOURCES = 2, POST_FINISH = 4, POST_ALL = 7,
       POST_SKIP_CELL_DETECTORS = 8 };

// Which whole-step kernel a run may use.
//   FUSED_CELL  : step_fused, one lane per cell re-deriving its neighbours' new E -- wins on
//                 small 2-D grids, where the launch count decides the step time;
//   FUSED_MARCH : step_march, workgroups marching along x -- wins on large 3-D grids, where
//                 it moves 12 instead of 18 values per cell and step.
enum FusedKind { FUSED_NONE = 0, FUSED_CELL, FUSED_MARCH };

enum { ST_E_FREE = 1, ST_MARCH_A = 2, ST_E_GHOST = 4, ST_MARCH_B = 8, ST_H_FREE = 16, ST_H_GHOST = 32,
       ST_ALL = 63 };

// ---- launch_half.hip --------------------------------------------------------------
int half_bulk(fdtdhip_plan* p, bool phaseE, int x0, int x1);
int half_post(fdtdhip_plan* p, bool phaseE, int64_t q, int parts = POST_ALL, int x0 = 0, int x1 = -1,
              int stage = -1);
int half_read_deferred_currents(fdtdhip_plan* p);
int half_scan_material(fdtdhip_plan* p, int w);
int half_curl(const void* in, void* out, int nx, int ny, int nz, int dtype, bool forward, void* stream);
int half_plane_energy(fdtdhip_plan* p, int axis, int index, void* outE, void* outH);

// ---- launch_whole.hip -------------------------------------------------------------
bool march_core(const fdtdhip_plan* p, int (&core)[6], int (&hole)[6]);
// shell_share: the largest share of the cells that the shell of the one-step march may hold
constexpr double MARCH_SHELL_SHARE = 1.0 / 6;
FusedKind fused_kind(const fdtdhip_plan* p, bool staged = false, double shell_share = MARCH_SHELL_SHARE);
int whole_fused_step(fdtdhip_plan* p, int64_t q, FusedKind kind);
// two steps per pass (yee_march2.hpp); inner_out (may be null): the box of the two-step kernel
bool whole_pair_possible(const fdtdhip_plan* p, int32_t inner_out[6] = nullptr, bool staged = false);
int whole_pair_stage(fdtdhip_plan* p, int64_t q, int stage);   // x-slabs: the pair in eight pieces
int whole_pair_step(fdtdhip_plan* p, int64_t q);
int whole_staged_step(fdtdhip_plan* p, int64_t q, int stage);
int march_slots(fdtdhip_plan* p, const void* kernel, int threads);
int march_chunk_planes(int planes, int tiles, int slots, int extra = 1);

// ---- launch_adjoint.hip -----------------------------------------------------------
int adj_run(fdtdhip_plan* p, int64_t nsteps, void* lamE, void* lamH, void* work);
int adj_run_noalloc(fdtdhip_plan* p, int64_t nsteps, void* lamE, void* lamH);
int adj_cells(fdtdhip_plan* p, bool phaseE, void* lamE, void* lamH, void* work);
int adj_gather(fdtdhip_plan* p, bool phaseE, void* lamE, void* lamH, void* work, const void* wghost);
int adj_post(fdtdhip_plan* p, bool phaseE, void* lamE, void* lamH, int stage);
int adj_stage_possible(fdtdhip_plan* p);
int adj_stage(fdtdhip_plan* p, int stage, void* lamE0, void* lamH0, void* lamE1, void* lamH1, void* work,
              int flip, const void* ghost);
int adj_pack(fdtdhip_plan* p, bool phaseE, void* lamE, void* lamH, void* out, int flip);

}  // namespace fdtdhip
