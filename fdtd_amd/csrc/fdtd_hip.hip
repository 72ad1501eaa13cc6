// libfdtd_hip: C ABI (include/fdtd_hip.h) over the HIP kernels of this directory.
//
// Host side of a plan: keeps the borrowed field pointers, the ordered boundary
// list, the object / source / detector tables, and turns one half step of the
// reference's Grid (reference fdtd/grid.py:275-325) into a short, fixed
// sequence of kernel launches on the plan's stream.
#include "../../include/fdtd_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <set>
#include <string>
#include <vector>

#include "plan.hpp"
#include "yee_generic.hpp"
#include "yee_tiled.hpp"
#include "yee_adjoint.hpp"
#include "yee_step.hpp"
#include "yee_march.hpp"
#include "yee_shell.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

const char* const kStaged =
    "a step taken in stages (fdtdhip_step_stage) is in progress: finish it with the remaining stages first";

const char* const kPoisoned =
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

}  // namespace

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
  std::vector<BoundaryRef> boundaries;
  std::vector<PmlH> pmls;
  std::vector<ObjH> objs;
  std::vector<SourceH> sources;
  std::vector<DetH> dets;
  void* obj_table = nullptr;  // device ObjK<T>[]
  size_t obj_table_bytes = 0;
  bool obj_dirty = false;
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
  void* adj_work = nullptr;   // second `work` array of the paired adjoint launches (owned)
  int64_t adj_work_bytes = 0;
  int64_t trace_cursor = -1;  // row of the curl traces the next E half step writes; -1: no traces
  std::vector<void*> owned;   // device allocations released by reset/destroy
  // optional kernel timing (fdtdhip_plan_enable_timing)
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  std::vector<int> ev_phase;  // what each used pair brackets (TIME_*)
  double ms[4] = {0, 0, 0, 0};
  int64_t launches[4] = {0, 0, 0, 0};
  size_t esize() const { return dtype == FDTDHIP_F32 ? 4 : 8; }
  int64_t cells() const { return (int64_t)nx * ny * nz; }
};

namespace {

using namespace fdtdhip;

inline unsigned blocks_for(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

int release_tables(fdtdhip_plan* p) {
  for (void* d : p->owned) (void)hipFree(d);
  p->owned.clear();
  if (p->obj_table) (void)hipFree(p->obj_table);
  p->obj_table = nullptr;
  p->obj_table_bytes = 0;
  p->trace_cursor = -1;
  p->pml_dirty = true;
  p->step_flipped = false;
  p->poisoned = false;
  p->stage_next = 0;
  p->boundaries.clear();
  p->pmls.clear();
  p->objs.clear();
  p->sources.clear();
  p->dets.clear();
  p->obj_dirty = false;
  p->mat_const[0] = p->mat_const[1] = false;
  p->detmap = nullptr;  // was in `owned`
  p->detmap_state = 0;
  p->adj_work = nullptr;  // was in `owned`
  p->adj_work_bytes = 0;
  return 0;
}

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

// Stream-ordered copy of `bytes` at `src` (host) to `dst` (device).  Without waiting for the
// stream, except inside the no-allocation entry point, which may not create the pinned
// buffers and copies out of the caller's memory instead.
int upload_table(fdtdhip_plan* p, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return 0;
  if (p->no_alloc) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    return 0;
  }
  StagingRing& r = p->staging;
  const int slot = r.next;
  r.next = (r.next + 1) % StagingRing::SLOTS;
  if (r.done[slot]) HIPCHK(hipEventSynchronize(r.done[slot]));
  else HIPCHK(hipEventCreateWithFlags(&r.done[slot], hipEventDisableTiming));
  if (r.cap[slot] < bytes) {
    if (r.host[slot]) HIPCHK(hipHostFree(r.host[slot]));
    r.host[slot] = nullptr;
    r.cap[slot] = 0;
    const size_t want = std::max<size_t>(bytes, 4096);
    HIPCHK(hipHostMalloc(&r.host[slot], want, hipHostMallocDefault));
    r.cap[slot] = want;
  }
  memcpy(r.host[slot], src, bytes);
  HIPCHK(hipMemcpyAsync(dst, r.host[slot], bytes, hipMemcpyHostToDevice, p->stream));
  HIPCHK(hipEventRecord(r.done[slot], p->stream));
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
  return 0;
}

// Which PML descriptor table a kernel reads: the two half steps in place, the E half
// of the fused full step in its two directions (first -> second psi_E buffer and
// back), and the adjoint half steps (psi -> adjoint arrays).
enum PmlTab { TAB_E = 0, TAB_H, TAB_E_AB, TAB_E_BA, TAB_ADJ_E, TAB_ADJ_H, TAB_COUNT };

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

const Strategy& strategy() {
  static const Strategy st = [] {
    Strategy s;
    if (const char* e = getenv("FDTDHIP_BULK")) {
      std::string v(e);
      s.chosen = true;
      v += "+";
      size_t pos = 0, nxt;
      while ((nxt = v.find('+', pos)) != std::string::npos) {
        const std::string t = v.substr(pos, nxt - pos);
        pos = nxt + 1;
        if (t == "generic") s.fast = false;
        else if (t == "band") { s.fast = false; s.band = true; }
        else if (t == "flat") s.flat = true;
        else if (t == "rows") s.flat = false;
        else if (t == "nont") s.ntl = s.nts = false;
        else if (t == "ntl") { s.ntl = true; s.nts = false; }
        else if (t == "nts") { s.nts = true; s.ntl = false; }
        else if (t == "nt") s.ntl = s.nts = true;
      }
    }
    if (const char* e = getenv("FDTDHIP_FORCE")) s.force = atoi(e) != 0;
    return s;
  }();
  return st;
}

int drain_timing(fdtdhip_plan* p) {
  for (size_t i = 0; i < p->ev_phase.size(); ++i) {
    float ms = 0;
    HIPCHK(hipEventSynchronize(p->ev_pool[i].second));
    HIPCHK(hipEventElapsedTime(&ms, p->ev_pool[i].first, p->ev_pool[i].second));
    p->ms[p->ev_phase[i]] += ms;
    p->launches[p->ev_phase[i]] += 1;
  }
  p->ev_phase.clear();
  return 0;
}

template <typename T>
int bulk_launch(fdtdhip_plan* p, bool phaseE, int x0, int x1);

// What a pair of timing events brackets: a bulk half-step launch (H, E), a whole-step kernel
// (step_march, step_march_box, step_fused), or the half-step launches over the shell that
// accompany the marching kernel on plans with PMLs, objects or open slab faces.
enum { TIME_H = 0, TIME_E = 1, TIME_WHOLE = 2, TIME_SHELL = 3 };

// First event of a timed span; `slot` names the pair for span_end.  No-ops unless timing is on.
int span_begin(fdtdhip_plan* p, size_t& slot) {
  slot = 0;
  if (!p->timing) return 0;
  if (p->ev_phase.size() >= 2048)
    if (int rc = drain_timing(p)) return rc;
  slot = p->ev_phase.size();
  if (slot >= p->ev_pool.size()) {
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    p->ev_pool.push_back({a, b});
  }
  HIPCHK(hipEventRecord(p->ev_pool[slot].first, p->stream));
  return 0;
}
int span_end(fdtdhip_plan* p, size_t slot, int what) {
  if (!p->timing) return 0;
  HIPCHK(hipEventRecord(p->ev_pool[slot].second, p->stream));
  p->ev_phase.push_back(what);
  return 0;
}

template <typename T>
int bulk(fdtdhip_plan* p, bool phaseE, int x0, int x1) {
  if (!p->timing || x0 == x1) return bulk_launch<T>(p, phaseE, x0, x1);
  size_t slot;
  if (int rc = span_begin(p, slot)) return rc;
  const int rc = bulk_launch<T>(p, phaseE, x0, x1);
  if (int rc2 = span_end(p, slot, phaseE ? TIME_E : TIME_H)) return rc2;
  return rc;
}

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

// Decide whether one material array can be replaced by a constant (see cell_material).
template <typename T>
int scan_material(fdtdhip_plan* p, int w) {
  p->mat_const[w] = false;
  const T* mat = (const T*)(w == 0 ? p->inv_eps : p->inv_mu);
  if (!mat) return 0;
  if (int rc = sync_obj_table<T>(p)) return rc;
  const int nobj = w == 0 ? (int)p->objs.size() : 0;
  // a cell outside every object box gives the background value
  const int64_t ncell = p->cells(), plane = (int64_t)p->ny * p->nz;
  int64_t probe = -1;
  for (int64_t cell = 0; cell < ncell && cell < 65536 && probe < 0; ++cell) {
    const int i = (int)(cell / plane), j = (int)((cell / p->nz) % p->ny), k = (int)(cell % p->nz);
    bool covered = false;
    for (int o = 0; o < nobj; ++o) {
      const int* b = p->objs[o].box;
      covered |= i >= b[0] && i < b[1] && j >= b[2] && j < b[3] && k >= b[4] && k < b[5];
    }
    if (!covered) probe = cell;
  }
  if (probe < 0) return 0;
  if (!p->scan_flag) HIPCHK(hipMalloc((void**)&p->scan_flag, sizeof(int)));
  T bg[3];
  int differs = 0;
  HIPCHK(hipMemcpyAsync(bg, mat + probe * 3, sizeof(bg), hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipMemsetAsync(p->scan_flag, 0, sizeof(int), p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));
  hipLaunchKernelGGL((material_scan<T>), dim3(blocks_for(ncell)), dim3(256), 0, p->stream, mat,
                     ncell, p->ny, p->nz, (const ObjK<T>*)p->obj_table, nobj, bg[0], bg[1], bg[2],
                     p->scan_flag);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(&differs, p->scan_flag, sizeof(int), hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));
  if (!differs) {
    p->mat_const[w] = true;
    for (int c = 0; c < 3; ++c) p->mat_bg[w][c] = (double)bg[c];
  }
  return 0;
}

template <typename T>
int bulk_launch(fdtdhip_plan* p, bool phaseE, int x0, int x1) {
  if (p->poisoned) return fail(-5, kPoisoned);
  if (p->stage_next != 0) return fail(-1, kStaged);
  if (!p->E || !p->H || !p->inv_eps || !p->inv_mu) return fail(-1, "fields not set");
  if (x0 < 0 || x1 > p->nx || x0 > x1) return fail(-1, "bad x range");
  if (x0 == x1) return 0;
  for (const ObjH& o : p->objs)
    if (o.kind == FDTDHIP_OBJ_ANISOTROPIC_MODULATED && !o.modw)
      return fail(-1, "modulated object without weights (fdtdhip_plan_set_object_modifier)");
  if (int rc = sync_obj_table<T>(p)) return rc;
  BulkArgs<T> a;
  a.F = (T*)(phaseE ? p->E : p->H);
  a.G = (const T*)(phaseE ? p->H : p->E);
  a.mat = (const T*)(phaseE ? p->inv_eps : p->inv_mu);
  a.ghost = (const T*)(phaseE ? p->H_below : p->E_above);
  a.nx = p->nx; a.ny = p->ny; a.nz = p->nz;
  a.x0 = x0; a.x1 = x1;
  a.sc = (T)p->sc;
  a.npml = (int)p->pmls.size();
  a.nobj = phaseE ? (int)p->objs.size() : 0;
  a.obj = (const ObjK<T>*)p->obj_table;
  a.trace_step = phaseE ? p->trace_cursor : -1;
  if (int rc = sync_pml_tables<T>(p)) return rc;
  a.pml = pml_table<T>(p, phaseE ? TAB_E : TAB_H);
  fill_core<T>(p, a);
  fill_material<T>(p, phaseE, a);

  const bool pml = a.npml > 0, obj = a.nobj > 0;
  const int64_t plane = (int64_t)p->ny * p->nz;
  const Strategy st = strategy();
  const dim3 block(256);

  // Measured on MI355X (512^3 and 256^3, tools/kbench, run-to-run spread about 3 %).
  // float32: a wave per z-row segment with rotated segments (see lane_cell),
  // non-temporal loads of the streamed arrays and non-temporal stores: 6.0 TB/s
  // algorithmic with material arrays, 5.8 TB/s of the smaller traffic with constant
  // materials (a plain 3-read-1-write float4 stream reaches 5.85 TB/s here).
  // float64: cells are 24 B, every load splits in a 16 B and an 8 B instruction that
  // touch the same lines and non-temporal lines are dropped between the two, so plain
  // loads; the store goes through LDS so that full lines leave the wave (store3_wave),
  // non-temporal; same row mapping (PMC: 11 % fetch excess against 57 % with the flat
  // mapping).  5.1 TB/s.
  // plans with energy-modulated objects (small research grids) run the one kernel that
  // carries their code
  bool modulated = false;
  for (const ObjH& o : p->objs) modulated |= o.kind == FDTDHIP_OBJ_ANISOTROPIC_MODULATED;
  if (modulated) {
    const dim3 mgrid(blocks_for((int64_t)(x1 - x0) * plane));
    if (phaseE) {
      if (pml) hipLaunchKernelGGL((bulk_E_generic<T, true, true, false, true>), mgrid, block, 0, p->stream, a);
      else hipLaunchKernelGGL((bulk_E_generic<T, false, true, false, true>), mgrid, block, 0, p->stream, a);
    } else {
      if (pml) hipLaunchKernelGGL((bulk_H_generic<T, true, false>), mgrid, block, 0, p->stream, a);
      else hipLaunchKernelGGL((bulk_H_generic<T, false, false>), mgrid, block, 0, p->stream, a);
    }
    HIPCHK(hipGetLastError());
    return 0;
  }
  Strategy use = st;
  if (!st.chosen) {
    use.fast = true;
    use.ntl = sizeof(T) == 4;
    use.nts = true;
    // a row segment is 64 lanes: rows whose last segment is mostly empty (nz = 96: a
    // quarter of the lanes idle, 68 against 79 Gcells/s) run the flat mapping
    const int segs = (p->nz + 63) / 64;
    const bool ragged = p->nz * 10 < segs * 64 * 9;
    use.flat = ragged;
  }
  // planes from 1024 cells on (at least four 256-cell blocks, the last one no emptier than a
  // fifth of the launch); 4000 x 4000 x 1 float32: 82 against 64 Gcells/s with the generic kernel
  if (use.fast && (st.force || plane >= 1024) && (x1 - x0) <= 65535 && plane < (1ll << 30)) {
    // flat: 256 consecutive cells per workgroup; rows: a wave per z-row segment
    const bool flat = use.flat || p->nz < 48 || (p->ny + 3) / 4 > 65535;
    const dim3 grid = flat ? dim3((unsigned)((plane + 255) / 256), x1 - x0, 1)
                           : dim3((p->nz + 63) / 64, (p->ny + 3) / 4, x1 - x0);
#define FAST(FLAT, NT, NTS)                                                                         \
  if (phaseE) {                                                                                     \
    if (pml && obj) hipLaunchKernelGGL((bulk_E_rows<T, true, true, FLAT, NT, NTS>), grid, block, 0, p->stream, a);   \
    else if (pml) hipLaunchKernelGGL((bulk_E_rows<T, true, false, FLAT, NT, NTS>), grid, block, 0, p->stream, a);    \
    else if (obj) hipLaunchKernelGGL((bulk_E_rows<T, false, true, FLAT, NT, NTS>), grid, block, 0, p->stream, a);    \
    else hipLaunchKernelGGL((bulk_E_rows<T, false, false, FLAT, NT, NTS>), grid, block, 0, p->stream, a);            \
  } else {                                                                                          \
    if (pml) hipLaunchKernelGGL((bulk_H_rows<T, true, FLAT, NT, NTS>), grid, block, 0, p->stream, a);  \
    else hipLaunchKernelGGL((bulk_H_rows<T, false, FLAT, NT, NTS>), grid, block, 0, p->stream, a);     \
  }
#define FAST2(FLAT)                                                     \
  if (use.ntl && use.nts) { FAST(FLAT, true, true) }                    \
  else if (use.ntl) { FAST(FLAT, true, false) }                         \
  else if (use.nts) { FAST(FLAT, false, true) }                         \
  else { FAST(FLAT, false, false) }
    if (flat) { FAST2(true) } else { FAST2(false) }
#undef FAST2
#undef FAST
    HIPCHK(hipGetLastError());
    return 0;
  }

  const int64_t ncell = (int64_t)(x1 - x0) * plane;
  const dim3 grid(blocks_for(ncell));
  // XCD banding needs whole 256-cell blocks per plane, a multiple of 8 of them
  const bool band = st.band && (plane % 2048 == 0);
#define GENERIC(BAND)                                                                               \
  if (phaseE) {                                                                                     \
    if (pml && obj) hipLaunchKernelGGL((bulk_E_generic<T, true, true, BAND>), grid, block, 0, p->stream, a);  \
    else if (pml) hipLaunchKernelGGL((bulk_E_generic<T, true, false, BAND>), grid, block, 0, p->stream, a);   \
    else if (obj) hipLaunchKernelGGL((bulk_E_generic<T, false, true, BAND>), grid, block, 0, p->stream, a);   \
    else hipLaunchKernelGGL((bulk_E_generic<T, false, false, BAND>), grid, block, 0, p->stream, a);           \
  } else {                                                                                          \
    if (pml) hipLaunchKernelGGL((bulk_H_generic<T, true, BAND>), grid, block, 0, p->stream, a);     \
    else hipLaunchKernelGGL((bulk_H_generic<T, false, BAND>), grid, block, 0, p->stream, a);        \
  }
  if (band) { GENERIC(true) } else { GENERIC(false) }
#undef GENERIC
  HIPCHK(hipGetLastError());
  return 0;
}

double waveform(const SourceH& s, int64_t q) {
  // reference fdtd/sources.py:99-110 and fdtd/waveforms.py:8-9, same operation order
  if (s.pulse) {
    const int64_t t1 = (int64_t)(2 * M_PI / (s.frequency * s.hanning_dt / s.cycle));
    if (q < t1) {
      const double f = s.frequency, t = (double)q * s.hanning_dt, n = s.cycle;
      return (1.0 / 2.0) * (1 - std::cos(f * t / n)) * (std::sin(f * t));
    }
    return 0.0;
  }
  return std::sin(2 * M_PI * (double)q / s.period + s.phase);
}

// The pieces of a half step that follow the bulk pass, selectable so that a run can
// interleave plane chunks of the two half steps: boundaries (periodic copies, PMLs
// registered after a periodic boundary), sources restricted to planes [x0, x1),
// and the end-of-half-step work (detectors, curl-trace cursor).
enum { POST_BOUNDARIES = 1, POST_SOURCES = 2, POST_FINISH = 4, POST_ALL = 7,
       POST_SKIP_CELL_DETECTORS = 8 };

// The E pass of a step filled one row of the objects' curl traces.
void advance_trace_cursor(fdtdhip_plan* p) {
  if (p->trace_cursor < 0) return;
  ++p->trace_cursor;
  for (const ObjH& o : p->objs)
    if (o.trace && p->trace_cursor >= o.trace_capacity) p->trace_cursor = -1;  // traces full
}

// One reading of a loop-current detector from the present H field.
template <typename T>
int read_current(fdtdhip_plan* p, DetH& d) {
  if (d.n == 0) { ++d.usedH; return 0; }
  if (!d.outH || d.usedH >= d.capacity) return fail(-3, "detector output capacity exhausted");
  T* out = (T*)d.outH + d.usedH * d.n;
  hipLaunchKernelGGL((detect_current<T>), dim3(blocks_for(d.n, 64)), dim3(64), 0, p->stream,
                     (const T*)p->H, (const T*)d.below, d.xyz, out, d.n, p->nx, p->ny, p->nz,
                     (T)d.spacing);
  HIPCHK(hipGetLastError());
  ++d.usedH;
  return 0;
}

template <typename T>
int post(fdtdhip_plan* p, bool phaseE, int64_t q, int parts = POST_ALL, int x0 = 0, int x1 = -1,
         int stage = -1) {
  if (p->poisoned) return fail(-5, kPoisoned);
  T* F = (T*)(phaseE ? p->E : p->H);
  const T* mat = (const T*)(phaseE ? p->inv_eps : p->inv_mu);
  if (!F || !mat) return fail(-1, "fields not set");
  if (x1 < 0) x1 = p->nx;
  const int64_t plane3 = (int64_t)p->ny * p->nz * 3;
  const int64_t elem_lo = x0 * plane3, elem_hi = x1 * plane3;
  // (the row of the curl traces moves on once per E half step: not in stage 0 of a pair)
  if ((parts & POST_FINISH) && phaseE && stage != 0) advance_trace_cursor(p);
  // stage 0 / 1: only what precedes / follows the x-periodic boundary whose copy crosses
  // ranks (the caller performs it in between); sources and the rest belong to stage 1
  size_t ext = p->boundaries.size();
  for (size_t n = 0; n < p->boundaries.size(); ++n)
    if (p->boundaries[n].kind == 1 && p->boundaries[n].idx == FDTDHIP_PERIODIC_X_EXTERNAL) {
      ext = n;
      break;
    }
  if (stage >= 0 && ext == p->boundaries.size()) return fail(-1, "no external periodic boundary");
  if (stage == 0) parts &= POST_BOUNDARIES;
  // boundaries that follow the first periodic one, in list order
  bool seen_periodic = false;
  for (size_t n = 0; n < p->boundaries.size(); ++n) {
    const BoundaryRef& b = p->boundaries[n];
    if (!(parts & POST_BOUNDARIES)) break;
    const bool active = stage < 0 || (stage == 0 ? n < ext : n > ext);
    if (b.kind == 1) {
      seen_periodic = true;
      if (!active || b.idx == FDTDHIP_PERIODIC_X_EXTERNAL) continue;
      const int axis = b.idx;
      const int n = axis == 0 ? p->nx : (axis == 1 ? p->ny : p->nz);
      if (n < 2) continue;  // copy of a plane onto itself
      const int64_t total = p->cells() / n * 3;
      const int dst = phaseE ? 0 : n - 1, src = phaseE ? n - 1 : 0;
      hipLaunchKernelGGL((periodic_copy<T>), dim3(blocks_for(total)), dim3(256), 0, p->stream, F,
                         p->nx, p->ny, p->nz, axis, dst, src);
      HIPCHK(hipGetLastError());
    } else if (seen_periodic && active) {
      const PmlH& h = p->pmls[b.idx];
      PmlK<T> k = make_pmlk<T>(p, h, phaseE);
      const int64_t slab = (int64_t)(h.axis == 0 ? h.t : p->nx) * k.sy * k.sz;
      hipLaunchKernelGGL((pml_apply_slab<T>), dim3(blocks_for(slab)), dim3(256), 0, p->stream, F,
                         mat, k, p->nx, p->ny, p->nz, (T)p->sc, phaseE ? T(1) : T(-1));
      HIPCHK(hipGetLastError());
    }
  }
  for (SourceH& s : p->sources) {
    if (!(parts & POST_SOURCES)) break;
    if (s.kind == FDTDHIP_SRC_SOFT_POINT) {
      if (!phaseE || s.nwave < 0) continue;
      if (s.det < 0 || s.det >= (int)p->dets.size() || p->dets[s.det].kind != FDTDHIP_DET_CURRENT)
        return fail(-1, "soft source without its current detector (fdtdhip_plan_link_soft_source)");
      const DetH& d = p->dets[s.det];
      // the newest reading: in the detector's current output buffer, else the one the caller named
      const T* cur = (const T*)s.prev_current;
      if (d.usedH > 0 && d.outH && d.n > 0) cur = (const T*)d.outH + (d.usedH - 1) * d.n;
      if (s.vin && s.vused >= s.vcap) return fail(-3, "soft source output capacity exhausted");
      hipLaunchKernelGGL((source_soft_point<T>), dim3(1), dim3(1), 0, p->stream, F, s.elem0, s.wave,
                         s.nwave, q, s.impedance, s.spacing, cur,
                         s.vin ? (T*)s.vin + s.vused : nullptr, s.vout ? (T*)s.vout + s.vused : nullptr);
      HIPCHK(hipGetLastError());
      if (s.vin) ++s.vused;
      continue;
    }
    if (s.kind == FDTDHIP_SRC_MODULATED_BOX) {
      if (!phaseE || !s.modP) continue;  // unseeded: no action (reference fdtd/sources.py:583-585)
      const int64_t row = q - s.q0;
      if (row < 0 || row >= s.osc_rows) return fail(-1, "modulated source: step outside its oscillator table");
      const int64_t total =
          (int64_t)(s.box[1] - s.box[0]) * (s.box[3] - s.box[2]) * (s.box[5] - s.box[4]);
      if (total <= 0) continue;
      T* et = s.etrace ? (T*)s.etrace + row * total * 2 : nullptr;
      hipLaunchKernelGGL((source_modulated<T>), dim3(blocks_for(total)), dim3(256), 0, p->stream,
                         (T*)p->E, (const T*)p->H, p->ny, p->nz, s.box[0], s.box[1], s.box[2],
                         s.box[3], s.box[4], s.box[5], s.compE, (const T*)s.modP,
                         (const T*)s.osc + row * s.channels, (const T*)s.modw, s.channels, et);
      HIPCHK(hipGetLastError());
      continue;
    }
    const double w = waveform(s, q);
    if (s.kind == FDTDHIP_SRC_ADD_POINTS) {
      if (!phaseE || s.n == 0) continue;
      if (s.elem_max < elem_lo || s.elem_min >= elem_hi) continue;  // no point in these planes
      hipLaunchKernelGGL((source_add_points<T>), dim3(s.serial ? 1 : blocks_for(s.n, 64)),
                         dim3(64), 0, p->stream, F, s.elem, s.profile, w, s.n, s.serial, elem_lo,
                         elem_hi);
      HIPCHK(hipGetLastError());
    } else {
      const int64_t total =
          (int64_t)(s.box[1] - s.box[0]) * (s.box[3] - s.box[2]) * (s.box[5] - s.box[4]);
      if (total <= 0) continue;
      hipLaunchKernelGGL((source_set_box<T>), dim3(blocks_for(total)), dim3(256), 0, p->stream, F,
                         p->ny, p->nz, s.box[0], s.box[1], s.box[2], s.box[3], s.box[4], s.box[5],
                         phaseE ? s.compE : s.compH, (T)(s.amplitude * w));
      HIPCHK(hipGetLastError());
    }
  }
  // cell-list detectors share launches (up to MAX_DET_BATCH per launch); reading a field
  // does not change it, so their order among the detectors does not matter
  DetBatch<T> batch;
  batch.count = 0;
  batch.first[0] = 0;
  auto flush = [&]() -> int {
    if (batch.count == 0) return 0;
    hipLaunchKernelGGL((detect_cells_batch<T>), dim3(blocks_for(batch.first[batch.count])),
                       dim3(256), 0, p->stream, F, batch);
    batch.count = 0;
    HIPCHK(hipGetLastError());
    return 0;
  };
  for (DetH& d : p->dets) {
    if (!(parts & POST_FINISH)) break;
    if (d.kind == FDTDHIP_DET_CELLS && (parts & POST_SKIP_CELL_DETECTORS)) continue;
    if (d.kind == FDTDHIP_DET_CELLS) {
      int64_t& used = phaseE ? d.usedE : d.usedH;
      char* base = phaseE ? d.outE : d.outH;
      if (d.n == 0) { ++used; continue; }
      if (!base || used >= d.capacity) {
        if (int rc = flush()) return rc;
        return fail(-3, "detector output capacity exhausted");
      }
      batch.cell[batch.count] = d.cell;
      batch.out[batch.count] = (T*)base + used * d.n * 3;
      batch.first[batch.count + 1] = batch.first[batch.count] + d.n * 3;
      ++used;
      if (++batch.count == MAX_DET_BATCH)
        if (int rc = flush()) return rc;
    } else if (!phaseE && !d.below) {
      if (int rc = read_current<T>(p, d)) { (void)flush(); return rc; }
    }
  }
  return flush();
}

// The current detectors that wait for the plane under the slab (fdtdhip_plan_set_detector_below).
template <typename T>
int read_deferred_currents(fdtdhip_plan* p) {
  if (!p->H) return fail(-1, "fields not set");
  for (DetH& d : p->dets)
    if (d.kind == FDTDHIP_DET_CURRENT && d.below)
      if (int rc = read_current<T>(p, d)) return rc;
  HIPCHK(hipGetLastError());
  return 0;
}

template <typename T>
int curl_launch(const void* in, void* out, int nx, int ny, int nz, bool forward, void* stream) {
  const int64_t n = (int64_t)nx * ny * nz;
  if (n == 0) return 0;
  if (forward)
    hipLaunchKernelGGL((curl_only<T, true>), dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)in, (T*)out, nx, ny, nz);
  else
    hipLaunchKernelGGL((curl_only<T, false>), dim3(blocks_for(n)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)in, (T*)out, nx, ny, nz);
  HIPCHK(hipGetLastError());
  return 0;
}

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

// Index of the periodic boundary along x whose two planes live on different ranks, or the
// number of boundaries.
size_t external_periodic(const fdtdhip_plan* p) {
  for (size_t b = 0; b < p->boundaries.size(); ++b)
    if (p->boundaries[b].kind == 1 && p->boundaries[b].idx == FDTDHIP_PERIODIC_X_EXTERNAL) return b;
  return p->boundaries.size();
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

// ---- fused full step (yee_step.hpp) -------------------------------------------------
// Usable when nothing has to happen between the two half steps that the fused
// kernel cannot do itself: no periodic boundary, no hard plane source, no x-slab
// halo, no curl trace being recorded, at most MAX_STEP_SOURCES additive source points
// without repeated cells, and the caller has provided the second buffers.
constexpr int64_t FUSED_AUTO_MAX_CELLS = 1 << 17;

// Which whole-step kernel a run may use.
//   FUSED_CELL  : step_fused, one lane per cell re-deriving its neighbours' new E -- wins on
//                 small 2-D grids, where the launch count decides the step time;
//   FUSED_MARCH : step_march, workgroups marching along x -- wins on large 3-D grids, where
//                 it moves 12 instead of 18 values per cell and step.
enum FusedKind { FUSED_NONE = 0, FUSED_CELL, FUSED_MARCH };
#ifndef FDTDHIP_MARCH_TY32
#define FDTDHIP_MARCH_TY32 4   // rows per workgroup of the marching kernel in float32 (experiments: -D)
#endif
#ifndef FDTDHIP_MARCH_TY64
#define FDTDHIP_MARCH_TY64 4   // the same in float64
#endif
constexpr int64_t MARCH_SHELL_SHARE = 6;   // shell cells at most a sixth of the grid

// Plans with PML faces take the marching step over the box no face reaches, with half steps
// of their own for the shell around it (yee_shell.hpp).  That needs a box.
// `hole`: the box that holds every object, clipped to the core (empty without objects): its E
// half step belongs to the shell as well.
bool march_core(const fdtdhip_plan* p, int (&core)[6], int (&hole)[6]) {
  BulkArgs<float> b;
  fill_core<float>(p, b);
  for (int n = 0; n < 6; ++n) core[n] = b.core[n];
  // on an x-slab the plane that reads the neighbour's H (E half step of plane 0) or E (H half
  // step of the last plane) waits for the halo exchange: it belongs to the shell
  if (p->H_below) core[0] = std::max(core[0], 1);
  if (p->E_above) core[1] = std::min(core[1], p->nx - 1);
  for (int ax = 0; ax < 3; ++ax) {
    if (core[2 * ax + 1] - core[2 * ax] < 1) return false;
    hole[2 * ax] = std::min(std::max(b.hull[2 * ax], core[2 * ax]), core[2 * ax + 1]);
    hole[2 * ax + 1] = std::min(std::max(b.hull[2 * ax + 1], hole[2 * ax]), core[2 * ax + 1]);
  }
  if (p->objs.empty() || hole[1] == hole[0] || hole[3] == hole[2] || hole[5] == hole[4])
    for (int n = 0; n < 6; ++n) hole[n] = 0;
  return true;
}

// Whether the marching kernel itself evaluates the E half step of the cell of element `elem`.
bool march_owns(const fdtdhip_plan* p, const int (&core)[6], const int (&hole)[6], int64_t elem) {
  const int64_t plane = (int64_t)p->ny * p->nz, cell = elem / 3;
  const int i = (int)(cell / plane), j = (int)((cell % plane) / p->nz), k = (int)(cell % p->nz);
  if (i < core[0] || i >= core[1] || j < core[2] || j >= core[3] || k < core[4] || k >= core[5])
    return false;
  return !(i >= hole[0] && i < hole[1] && j >= hole[2] && j < hole[3] && k >= hole[4] && k < hole[5]);
}

// `staged`: asked by fdtdhip_step_stage, the whole step in pieces around the halo exchanges
// of an x-slab -- the marching step or nothing.
FusedKind fused_kind(const fdtdhip_plan* p, bool staged = false) {
  // FDTDHIP_FUSED=0 never, =1 step_fused whenever possible, =2 step_march whenever possible
  // (tests).  By default step_fused only where it was measured faster than two half-step
  // launches (MI355X, PML on every face, tools/mid_cost.py): 2-D grids up to about 420 x 420
  // cells (161 x 97: 9.0 against 11.4 us per step, 360^2: 12.7 / 14.1, 420^2: 15.4 / 15.7,
  // 512^2: 17.7 / 17.1; in 3-D the lane's four E evaluations with six faces cost more than
  // the second launch at every size tried, and on large grids the kernel is L1-bound), and
  // step_march on 3-D grids without PMLs and objects, with z rows of 128 to 1024 cells and enough rows and
  // planes to fill the machine with workgroups (512^3 float32: 1.25 against 1.66 ms per step,
  // 256^3: 0.187 against 0.21, 256 x 1024 x 1024: 2.67 against 3.3; float64 512^3: 3.05 / 3.34).
  static const int mode = [] {
    const char* e = getenv("FDTDHIP_FUSED");
    return e ? atoi(e) : -1;
  }();
  if (mode == 0 || !p->E_alt || !p->H_alt) return FUSED_NONE;
  for (const ObjH& o : p->objs)
    if (o.kind == FDTDHIP_OBJ_ANISOTROPIC_MODULATED) return FUSED_NONE;  // only the generic E kernel has them
  const bool halo = p->H_below || p->E_above;
  if (halo && !staged) return FUSED_NONE;
  if (staged && p->trace_cursor >= 0) return FUSED_NONE;   // recorded runs on slabs: half steps
  for (const DetH& d : p->dets)
    if (staged && d.below) return FUSED_NONE;   // read between the post pieces and a plane exchange
  for (const BoundaryRef& b : p->boundaries)
    if (b.kind == 1) return FUSED_NONE;
  for (const PmlH& h : p->pmls)
    if (!h.psiE_alt) return FUSED_NONE;
  int64_t points = 0;
  for (const SourceH& s : p->sources) {
    if (s.kind != FDTDHIP_SRC_ADD_POINTS || s.serial) return FUSED_NONE;
    points += s.n;
  }
  if (points > MAX_STEP_SOURCES) return FUSED_NONE;
  const bool cell_ok = p->nx <= 65535 && (int64_t)p->ny * p->nz < (1ll << 30) &&
                       p->cells() * 3 < (1ll << 31);   // the kernel's element offsets are 32-bit
  int core[6], hole[6] = {0, 0, 0, 0, 0, 0};
  const bool whole_grid = p->pmls.empty() && p->objs.empty() && !halo;   // step_march, no shell
  const bool march_ok = (whole_grid || march_core(p, core, hole)) &&
                        p->nz <= 64 * MARCH_MAX_WAVES && p->nx / 16 <= 65535;
  // the cells that take half steps of their own next to the marching kernel
  const int64_t shell_cells = whole_grid || !march_ok ? 0 :
      p->cells() - (int64_t)(core[1] - core[0]) * (core[3] - core[2]) * (core[5] - core[4]) +
      (int64_t)(hole[1] - hole[0]) * (hole[3] - hole[2]) * (hole[5] - hole[4]);
  if (mode == 1) return cell_ok && !staged ? FUSED_CELL : FUSED_NONE;
  if (mode == 2) return march_ok ? FUSED_MARCH : FUSED_NONE;
  const int thin = std::min(p->nx, std::min(p->ny, p->nz));
  if (cell_ok && !staged && p->cells() <= FUSED_AUTO_MAX_CELLS && thin == 1) return FUSED_CELL;
  // (with material arrays to read, the shell's half steps cost more than the marching kernel
  // saves: 512^3 float32, six faces and two 64^3 boxes, 2.44 against 2.26 ms per step)
  if (!(p->mat_const[0] && p->mat_const[1]) && !(p->pmls.empty() && p->objs.empty())) return FUSED_NONE;
  // (PML faces and objects: the shell and the box of the objects take half steps of their
  // own at a third of the marching kernel's rate per cell -- worth it while they are few)
  // (256^3 with six ten-cell faces, 22 % of the cells: float64 0.489 against 0.508 ms per step
  // in a bare loop and no difference in the benchmark, float32 0.256 against 0.235)
  if (march_ok && p->nz >= 128 && p->ny >= 64 && p->nx >= 64 && p->cells() >= (1ll << 22) &&
      shell_cells * MARCH_SHELL_SHARE <= p->cells())
    return FUSED_MARCH;
  return FUSED_NONE;
}

// Room for `n` more readings of every detector, checked before a fused pair of steps
// enqueues anything (a failure between its two steps would strand the state in the second
// buffers).
int capacity_check(const fdtdhip_plan* p, int64_t n) {
  for (const DetH& d : p->dets) {
    if (d.n == 0) continue;
    const bool cells = d.kind == FDTDHIP_DET_CELLS;
    if (!cells && d.below) continue;   // read on request (fdtdhip_read_deferred_currents)
    if (!d.outH || (cells && !d.outE) || d.usedH + n > d.capacity ||
        (cells && d.usedE + n > d.capacity))
      return fail(-3, "detector output capacity exhausted");
  }
  return 0;
}

// Build the detector map of the fused step once per plan layout.
int build_detmap(fdtdhip_plan* p) {
  p->detmap_state = -1;
  p->detmap = nullptr;   // an earlier map stays in `owned` until the plan is reset
  int cells_dets = 0;
  for (const DetH& d : p->dets)
    if (d.kind == FDTDHIP_DET_CELLS && d.n > 0) ++cells_dets;
  if (cells_dets == 0 || cells_dets > MAX_STEP_DETECTORS) return 0;
  std::vector<int32_t> map((size_t)p->cells(), -1);
  int slot = 0;
  for (const DetH& d : p->dets) {
    if (d.kind != FDTDHIP_DET_CELLS || d.n == 0) continue;
    if (d.n >= (1 << 24)) return 0;
    for (int64_t n = 0; n < d.n; ++n) {
      int32_t& m = map[(size_t)d.cell_host[n]];
      if (m >= 0) return 0;  // a cell read twice: leave it to the detector kernels
      m = (slot << 24) | (int32_t)n;
    }
    ++slot;
  }
  HIPCHK(hipMalloc((void**)&p->detmap, sizeof(int32_t) * map.size()));
  p->owned.push_back(p->detmap);
  HIPCHK(hipMemcpy(p->detmap, map.data(), sizeof(int32_t) * map.size(), hipMemcpyHostToDevice));
  p->detmap_state = 1;
  return 0;
}

// Workgroups of `kernel` (blocks of `threads`) that the GPU of the plan holds at once.
int march_slots(fdtdhip_plan* p, const void* kernel, int threads) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> per_cu;
  static std::map<int, int> cus;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_pair(kernel, threads);
  if (!per_cu.count(key)) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, 0) != hipSuccess || n < 1) n = 2;
    per_cu[key] = n;
  }
  if (!cus.count(p->device)) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, p->device) != hipSuccess || n < 1) n = 256;
    cus[p->device] = n;
  }
  return per_cu[key] * cus[p->device];
}

// Planes per chunk of the marching kernels.  A workgroup takes one tile of rows through one
// chunk of planes; all of them should keep the machine busy in whole rounds: 960 workgroups on
// 768 places (256^3 float64 with six faces) left three quarters of the CUs idle for a fifth of
// the launch (340 -> 323 us).  Among the chunk lengths of 16 to 128 planes the one with the
// fullest rounds wins, counting the plane every chunk evaluates past its end, and a single
// round counting 3 % less than several (512^3 float32: 4 chunks of 128 planes in one round
// 1.215 ms, 8 chunks of 64 in two rounds 1.183).
int march_chunk_planes(int planes, int tiles, int slots) {
  static const int forced_lx = [] {
    const char* e = getenv("FDTDHIP_MARCH_LX");
    return e ? atoi(e) : 0;
  }();
  if (forced_lx > 0) return forced_lx;
  int best_lx = std::min(planes, 128);
  double best = -1;
  for (int chunks = 1; chunks <= std::max(1, planes / 16); ++chunks) {
    const int lx = (planes + chunks - 1) / chunks;
    if (lx > 128) continue;
    const int64_t wgs = (int64_t)tiles * ((planes + lx - 1) / lx);
    const int64_t rounds = (wgs + slots - 1) / slots;
    const double score = (double)wgs / (double)(rounds * slots) * lx / (lx + 1.0) * (rounds >= 2 ? 1.0 : 0.97);
    if (score > best + 1e-9) { best = score; best_lx = lx; }
  }
  return std::max(best_lx, 1);
}

// The marching step (yee_march.hpp) over the whole grid: plans without PMLs, objects and
// open slab faces.  (The kernel carries no PML code: with the face code evaluated row after
// row every workgroup waited, plane by plane, for its two waves on the z faces to walk their
// chain of memory-variable round trips -- 512^3 float32, six faces: 4.6 ms per step with 2
// rows, 6.2 ms with 3, against 1.81 ms for two half-step launches.  Plans with faces split
// off a shell instead, see launch_march_shell.)
template <typename T, int TY, int MW>
int launch_march_tiles(fdtdhip_plan* p, const StepArgs<T>& a, bool obj) {
  MarchArgs<T> m;
  m.s = a;
  m.chunk0 = 0;
  const int tiles = (p->ny + TY - 1) / TY;
  const int waves = (p->nz + 63) / 64;
  const bool constants = p->mat_const[0] && p->mat_const[1];
  // (a kernel limited to 8 waves per workgroup may use 256 registers, one for z rows of up to
  // 1024 cells 128)
  void (*kernel)(MarchArgs<T>) = obj ? step_march<T, TY, false, true, false, MW>
                                 : constants ? step_march<T, TY, false, false, true, MW>
                                             : step_march<T, TY, false, false, false, MW>;
  const int lx = m.lx = march_chunk_planes(p->nx, tiles, march_slots(p, (const void*)kernel, 64 * waves));
  // tiles dealt to the XCDs in bands (see step_march); FDTDHIP_MARCH_BAND=0: plain order
  static const bool banded = [] {
    const char* e = getenv("FDTDHIP_MARCH_BAND");
    return !e || atoi(e) != 0;
  }();
  m.band = banded && tiles >= 16 ? (tiles + 7) / 8 : 0;
  const dim3 grid(m.band ? 8 * m.band : tiles, (p->nx + lx - 1) / lx, 1), block(64 * waves);
  hipLaunchKernelGGL(kernel, grid, block, 0, p->stream, m);
  return 0;
}

// Rows per workgroup and workgroup size of the marching kernels.  float32: 4 rows.  float64:
// 4 rows in workgroups of up to 8 waves (z rows of up to 512 cells: 160-168 registers, 3 waves
// per SIMD; 512^3 vacuum 2.43 ms per step against 2.53 with 2 rows -- 1, 3, 6 and 8 rows:
// 2.98, 2.92, 2.76, 2.74), 2 rows in the 16-wave workgroups of longer z rows, which may use
// 128 registers and would spill with 4.  (float32 with 2, 3, 5, 6, 8 rows: 1.22, 1.45, 1.38,
// 1.36, 1.18 against 1.19 ms with 4.)
template <typename T>
int launch_march(fdtdhip_plan* p, const StepArgs<T>& a, bool obj) {
  if ((p->nz + 63) / 64 <= 8)
    return launch_march_tiles<T, sizeof(T) == 4 ? FDTDHIP_MARCH_TY32 : FDTDHIP_MARCH_TY64, 8>(p, a, obj);
  return launch_march_tiles<T, sizeof(T) == 4 ? FDTDHIP_MARCH_TY32 : 2, MARCH_MAX_WAVES>(p, a, obj);
}

// The marching step of a plan with PML faces, objects or open slab faces, in pieces: E half
// step of the shell, the marching kernel over the core, H half step of the shell
// (yee_shell.hpp).  On an x-slab two planes wait for a neighbour's plane -- plane 0 in the E
// half step, the last plane in the H half step -- and the marching kernel comes in two
// launches over halves of its chunks of planes, so that a caller can run
//   E of the shell without plane 0, first half of the core      while H_below arrives,
//   E of plane 0                                                 (then sends its new E),
//   second half of the core, H of the shell without the last plane   while E_above arrives,
//   H of the last plane.
// The marching kernel reads the new E of plane 0 nowhere (its lowest plane is plane 1 at
// least, whose H half step reads plane 2), so the order is that of two half steps for every
// operand.  Without open faces all pieces are enqueued at once, the core in one launch.
enum { ST_E_FREE = 1, ST_MARCH_A = 2, ST_E_GHOST = 4, ST_MARCH_B = 8, ST_H_FREE = 16, ST_H_GHOST = 32,
       ST_ALL = 63 };

template <typename T, int TY, int MW>
int launch_march_shell_tiles(fdtdhip_plan* p, const StepArgs<T>& a, bool obj, int stages) {
  int c[6], hole[6];
  if (!march_core(p, c, hole)) return fail(-1, "internal: no core box for the marching step");
  const int nx = p->nx, ny = p->ny, nz = p->nz;
  const int e_lo = p->H_below ? 1 : 0;        // E half step: plane 0 waits for H_below
  const int h_hi = p->E_above ? nx - 1 : nx;  // H half step: the last plane waits for E_above
  // the shell as disjoint boxes: x slabs over everything, y slabs between them, z slabs
  // inside; and the box of the objects, for the E half step only
  auto table = [&](std::initializer_list<ShellBox> boxes) {
    Shell sh;
    sh.count = 0;
    sh.first[0] = 0;
    for (const ShellBox& b : boxes) {
      const int64_t n = (int64_t)std::max(b.i1 - b.i0, 0) * std::max(b.j1 - b.j0, 0) * std::max(b.k1 - b.k0, 0);
      if (n == 0) continue;
      sh.box[sh.count] = b;
      sh.first[sh.count + 1] = sh.first[sh.count] + blocks_for(n);
      ++sh.count;
    }
    for (int n = sh.count; n < MAX_SHELL_BOXES; ++n) {
      sh.box[n] = ShellBox{0, 0, 0, 0, 0, 0};
      sh.first[n + 1] = sh.first[sh.count];
    }
    return sh;
  };
  const ShellBox ylow{c[0], c[1], 0, c[2], 0, nz}, yhigh{c[0], c[1], c[3], ny, 0, nz},
      zlow{c[0], c[1], c[2], c[3], 0, c[4]}, zhigh{c[0], c[1], c[2], c[3], c[5], nz};
  const bool pml = !p->pmls.empty();
  size_t slot;
  auto shell_e = [&](const Shell& sh) -> int {
    if (!sh.count) return 0;
    if (int rc = span_begin(p, slot)) return rc;
    const dim3 grid((unsigned)sh.first[sh.count]), block(256);
    if (pml && obj) hipLaunchKernelGGL((shell_E<T, true, true>), grid, block, 0, p->stream, a.e, a.E1, sh);
    else if (pml) hipLaunchKernelGGL((shell_E<T, true, false>), grid, block, 0, p->stream, a.e, a.E1, sh);
    else if (obj) hipLaunchKernelGGL((shell_E<T, false, true>), grid, block, 0, p->stream, a.e, a.E1, sh);
    else hipLaunchKernelGGL((shell_E<T, false, false>), grid, block, 0, p->stream, a.e, a.E1, sh);
    HIPCHK(hipGetLastError());
    return span_end(p, slot, TIME_SHELL);
  };
  auto shell_h = [&](const Shell& sh) -> int {
    if (!sh.count) return 0;
    if (int rc = span_begin(p, slot)) return rc;
    BulkArgs<T> h = a.h;
    h.G = a.E1;     // the H half step is driven by the new E
    const dim3 grid((unsigned)sh.first[sh.count]), block(256);
    if (pml) hipLaunchKernelGGL((shell_H<T, true>), grid, block, 0, p->stream, h, a.H1, sh);
    else hipLaunchKernelGGL((shell_H<T, false>), grid, block, 0, p->stream, h, a.H1, sh);
    HIPCHK(hipGetLastError());
    return span_end(p, slot, TIME_SHELL);
  };
  // source points: the marching kernel adds those on cells whose E half step it evaluates,
  // the others are added to the shell's new E once that is written (plane 0 of an open slab
  // apart from the rest)
  MarchArgs<T> m;
  m.s = a;
  ShellSources<T> ss_free, ss_ghost;
  ss_free.n = ss_ghost.n = m.s.nsrc = 0;
  const int64_t plane3 = (int64_t)ny * nz * 3;
  for (int n = 0; n < a.nsrc; ++n) {
    if (march_owns(p, c, hole, a.src_elem[n])) {
      m.s.src_elem[m.s.nsrc] = a.src_elem[n];
      m.s.src_value[m.s.nsrc++] = a.src_value[n];
    } else {
      ShellSources<T>& ss = e_lo && a.src_elem[n] < plane3 ? ss_ghost : ss_free;
      ss.elem[ss.n] = a.src_elem[n];
      ss.value[ss.n++] = a.src_value[n];
    }
  }
  if (stages & ST_E_FREE) {
    if (int rc = shell_e(table({{e_lo, c[0], 0, ny, 0, nz}, {c[1], nx, 0, ny, 0, nz}, ylow, yhigh, zlow, zhigh,
                                {hole[0], hole[1], hole[2], hole[3], hole[4], hole[5]}})))
      return rc;
    if (ss_free.n) hipLaunchKernelGGL((shell_sources<T>), dim3(1), dim3(64), 0, p->stream, a.E1, ss_free);
  }
  if (stages & ST_E_GHOST) {
    if (int rc = shell_e(table({{0, e_lo, 0, ny, 0, nz}}))) return rc;
    if (ss_ghost.n) hipLaunchKernelGGL((shell_sources<T>), dim3(1), dim3(64), 0, p->stream, a.E1, ss_ghost);
  }
  if (stages & (ST_MARCH_A | ST_MARCH_B)) {
    for (int n = 0; n < 6; ++n) { m.box[n] = c[n]; m.hole[n] = hole[n]; }
    const bool holed = hole[1] > hole[0];
    const int tiles = (c[3] - c[2] + TY - 1) / TY, planes = c[1] - c[0];
    const int waves = (nz + 63) / 64;
    const bool constants = p->mat_const[0] && p->mat_const[1];
    void (*kernel)(MarchArgs<T>) = holed ? (constants ? step_march_box<T, TY, true, true, MW>
                                                      : step_march_box<T, TY, true, false, MW>)
                                         : (constants ? step_march_box<T, TY, false, true, MW>
                                                      : step_march_box<T, TY, false, false, MW>);
    const bool both = (stages & ST_MARCH_A) && (stages & ST_MARCH_B);
    // (in two launches: each half of the planes should fill its rounds)
    const int lx = m.lx = march_chunk_planes(both ? planes : (planes + 1) / 2, tiles,
                                             march_slots(p, (const void*)kernel, 64 * waves));
    const int nchunks = (planes + lx - 1) / lx, half = nchunks / 2;
    m.band = tiles >= 16 ? (tiles + 7) / 8 : 0;
    m.chunk0 = both || (stages & ST_MARCH_A) ? 0 : half;
    const int count = both ? nchunks : ((stages & ST_MARCH_A) ? half : nchunks - half);
    if (count > 0) {
      const dim3 grid(m.band ? 8 * m.band : tiles, count, 1), block(64 * waves);
      if (int rc = span_begin(p, slot)) return rc;
      hipLaunchKernelGGL(kernel, grid, block, 0, p->stream, m);
      HIPCHK(hipGetLastError());
      if (int rc = span_end(p, slot, TIME_WHOLE)) return rc;
    }
  }
  if (stages & ST_H_FREE)
    if (int rc = shell_h(table({{0, c[0], 0, ny, 0, nz}, {c[1], h_hi, 0, ny, 0, nz}, ylow, yhigh, zlow, zhigh})))
      return rc;
  if (stages & ST_H_GHOST)
    if (int rc = shell_h(table({{h_hi, nx, 0, ny, 0, nz}}))) return rc;
  HIPCHK(hipGetLastError());
  return 0;
}

template <typename T>
int launch_march_shell(fdtdhip_plan* p, const StepArgs<T>& a, bool obj, int stages) {
  if ((p->nz + 63) / 64 <= 8)
    return launch_march_shell_tiles<T, sizeof(T) == 4 ? FDTDHIP_MARCH_TY32 : FDTDHIP_MARCH_TY64, 8>(p, a, obj, stages);
  return launch_march_shell_tiles<T, sizeof(T) == 4 ? FDTDHIP_MARCH_TY32 : 2, MARCH_MAX_WAVES>(p, a, obj, stages);
}

// Kernel arguments of a whole step at time index q (both half steps, the second buffers,
// the additive source points with their values).
template <typename T>
int step_args(fdtdhip_plan* p, int64_t q, bool march, StepArgs<T>& a) {
  if (int rc = sync_obj_table<T>(p)) return rc;
  for (int phase = 0; phase < 2; ++phase) {
    const bool phaseE = phase == 0;
    BulkArgs<T>& b = phaseE ? a.e : a.h;
    b.F = (T*)(phaseE ? p->E : p->H);
    b.G = (const T*)(phaseE ? p->H : p->E);
    b.mat = (const T*)(phaseE ? p->inv_eps : p->inv_mu);
    b.ghost = (const T*)(phaseE ? p->H_below : p->E_above);   // (open slab faces: marching step only)
    b.nx = p->nx; b.ny = p->ny; b.nz = p->nz;
    b.x0 = 0; b.x1 = p->nx;
    b.sc = (T)p->sc;
    b.npml = (int)p->pmls.size();
    b.nobj = phaseE ? (int)p->objs.size() : 0;
    b.obj = (const ObjK<T>*)p->obj_table;
    b.trace_step = phaseE ? p->trace_cursor : -1;   // the owner lane records the curl
    fill_core<T>(p, b);
    fill_material<T>(p, phaseE, b);
  }
  if (int rc = sync_pml_tables<T>(p)) return rc;
  // p->step_flipped: the state currently lives in the second buffers
  // (the marching step advances the memory variables in place: its shell cells are evaluated once)
  a.e.pml = pml_table<T>(p, march ? TAB_E : (p->step_flipped ? TAB_E_BA : TAB_E_AB));
  a.h.pml = pml_table<T>(p, TAB_H);
  a.E1 = (T*)p->E_alt;
  a.H1 = (T*)p->H_alt;
  a.nsrc = 0;
  a.detmap = nullptr;
  for (int slot = 0; slot < MAX_STEP_DETECTORS; ++slot) a.detE[slot] = a.detH[slot] = nullptr;
  for (const SourceH& s : p->sources) {
    const double w = waveform(s, q);
    for (int64_t i = 0; i < s.n; ++i) {
      if (a.nsrc >= MAX_STEP_SOURCES) return fail(-1, "internal: too many source points for a whole step");
      a.src_elem[a.nsrc] = s.elem_host[i];
      a.src_value[a.nsrc] = (T)(s.profile_host[i] * w);
      ++a.nsrc;
    }
  }
  return 0;
}

// What follows the kernels of a whole step: the second buffers now hold the current state.
template <typename T>
int step_finish(fdtdhip_plan* p, int64_t q, bool cells_read) {
  std::swap(p->E, p->E_alt);
  std::swap(p->H, p->H_alt);
  p->step_flipped = !p->step_flipped;  // psi_E follows through the two fused PML tables
  if (cells_read) {
    // cell-list detectors were read by the kernel; empty ones and loop currents remain
    for (DetH& d : p->dets) {
      if (d.kind == FDTDHIP_DET_CELLS) { ++d.usedE; ++d.usedH; }
    }
    advance_trace_cursor(p);
    return post<T>(p, false, q, POST_FINISH | POST_SKIP_CELL_DETECTORS);
  }
  if (int rc = post<T>(p, true, q, POST_FINISH)) return rc;
  return post<T>(p, false, q, POST_FINISH);
}

template <typename T>
int fused_step(fdtdhip_plan* p, int64_t q, FusedKind kind) {
  const bool march = kind == FUSED_MARCH;   // reads no detectors itself
  if (!march && p->detmap_state == 0)
    if (int rc = build_detmap(p)) return rc;
  StepArgs<T> a;
  if (int rc = step_args<T>(p, q, march, a)) return rc;
  if (!march && p->detmap_state == 1) {
    int slot = 0;
    for (DetH& d : p->dets) {
      if (d.kind != FDTDHIP_DET_CELLS || d.n == 0) continue;
      if (!d.outE || !d.outH || d.usedE >= d.capacity || d.usedH >= d.capacity)
        return fail(-3, "detector output capacity exhausted");
      if (slot >= MAX_STEP_DETECTORS) return fail(-1, "detector map out of date");
      a.detE[slot] = (T*)d.outE + d.usedE * d.n * 3;
      a.detH[slot] = (T*)d.outH + d.usedH * d.n * 3;
      ++slot;
    }
    a.detmap = p->detmap;
  }
  const int64_t plane = (int64_t)p->ny * p->nz;
  const dim3 grid((unsigned)((plane + 255) / 256), p->nx, 1), block(256);
  const bool pml = a.e.npml > 0, obj = a.e.nobj > 0;
  if (march && (pml || obj)) {
    if (int rc = launch_march_shell<T>(p, a, obj, ST_ALL)) return rc;
  } else {
    size_t slot;
    if (int rc = span_begin(p, slot)) return rc;
    if (march) {
      if (int rc = launch_march<T>(p, a, obj)) return rc;
    } else if (pml && obj) hipLaunchKernelGGL((step_fused<T, true, true>), grid, block, 0, p->stream, a);
    else if (pml) hipLaunchKernelGGL((step_fused<T, true, false>), grid, block, 0, p->stream, a);
    else if (obj) hipLaunchKernelGGL((step_fused<T, false, true>), grid, block, 0, p->stream, a);
    else hipLaunchKernelGGL((step_fused<T, false, false>), grid, block, 0, p->stream, a);
    HIPCHK(hipGetLastError());
    if (int rc = span_end(p, slot, TIME_WHOLE)) return rc;
  }
  return step_finish<T>(p, q, a.detmap != nullptr);
}

// One piece of a whole step on an x-slab (fdtdhip_step_stage).
template <typename T>
int staged_step(fdtdhip_plan* p, int64_t q, int stage) {
  StepArgs<T> a;
  if (int rc = step_args<T>(p, q, true, a)) return rc;
  static const int pieces[4] = {ST_E_FREE | ST_MARCH_A, ST_E_GHOST, ST_MARCH_B | ST_H_FREE, ST_H_GHOST};
  if (int rc = launch_march_shell<T>(p, a, a.e.nobj > 0, pieces[stage])) return rc;
  return stage == 3 ? step_finish<T>(p, q, false) : 0;
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

// In host code this name is only a handle for hipGetSymbolAddress: every GPU holds its own
// copy of the array at its own address (noalloc_scratch below).
static __device__ __align__(256) unsigned char scratch[FDTDHIP_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int NOALLOC_MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};

// Device address of `scratch` on `device`, which must be the current device.  Looked up once
// per device.
int noalloc_scratch(int device, unsigned char** out) {
  if (device < 0 || device >= NOALLOC_MAX_DEVICES) return fail(-1, "device ordinal out of range");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void* ptr = nullptr;
    HIPCHK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(-2, "hipGetSymbolAddress returned no address for the scratch array");
    g_scratch_ptr[device] = (unsigned char*)ptr;
  }
  *out = g_scratch_ptr[device];
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
  if (int rc = noalloc_scratch(p->device, &base)) return rc;
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
  return rc;
}

#define DISPATCH(p, call_f32, call_f64) ((p)->dtype == FDTDHIP_F32 ? (call_f32) : (call_f64))

}  // namespace

extern "C" {

int fdtdhip_abi_version(void) { return FDTDHIP_ABI_VERSION; }

const char* fdtdhip_last_error(void) { return g_err.c_str(); }

int fdtdhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int fdtdhip_plan_create(fdtdhip_plan** out, int nx, int ny, int nz, int dtype, double courant,
                        int device) {
  if (!out) return fail(-1, "null output pointer");
  if (nx < 1 || ny < 1 || nz < 1) return fail(-1, "grid extents must be >= 1");
  if (dtype != FDTDHIP_F32 && dtype != FDTDHIP_F64) return fail(-1, "unknown dtype");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
    return fail(-4, "no HIP device available: the hip backend needs an AMD GPU");
  if (device < 0 || device >= n) return fail(-1, "device ordinal out of range");
  fdtdhip_plan* p = new fdtdhip_plan();
  p->nx = nx; p->ny = ny; p->nz = nz; p->dtype = dtype; p->device = device; p->sc = courant;
  *out = p;
  return 0;
}

int fdtdhip_plan_reset(fdtdhip_plan* p) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  (void)hipStreamSynchronize(p->stream);
  return release_tables(p);
}

int fdtdhip_plan_destroy(fdtdhip_plan* p) {
  if (!p) return 0;
  {
    DeviceGuard g(p->device);   // best effort: the host side is released in any case
    (void)hipStreamSynchronize(p->stream);
    release_tables(p);
    if (p->pml_tables) (void)hipFree(p->pml_tables);
    if (p->scan_flag) (void)hipFree(p->scan_flag);
    for (int n = 0; n < StagingRing::SLOTS; ++n) {
      if (p->staging.done[n]) (void)hipEventDestroy(p->staging.done[n]);
      if (p->staging.host[n]) (void)hipHostFree(p->staging.host[n]);
    }
    for (auto& e : p->ev_pool) {
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
  }
  delete p;
  return 0;
}

int fdtdhip_plan_set_stream(fdtdhip_plan* p, void* stream) {
  if (!p) return fail(-1, "null plan");
  p->stream = (hipStream_t)stream;
  return 0;
}

int fdtdhip_plan_set_fields(fdtdhip_plan* p, void* E, void* H, const void* inv_eps,
                            const void* inv_mu) {
  if (!p) return fail(-1, "null plan");
  if (!E || !H || !inv_eps || !inv_mu) return fail(-1, "null field pointer");
  // a new material array has not been scanned
  if (inv_eps != p->inv_eps) p->mat_const[0] = false;
  if (inv_mu != p->inv_mu) p->mat_const[1] = false;
  p->E = E; p->H = H; p->inv_eps = inv_eps; p->inv_mu = inv_mu;
  return 0;
}

int fdtdhip_plan_scan_materials(fdtdhip_plan* p, int enable) {
  if (!p) return fail(-1, "null plan");
  static const bool off = [] {
    const char* e = getenv("FDTDHIP_MATERIAL_SCAN");
    return e && atoi(e) == 0;
  }();
  p->mat_const[0] = p->mat_const[1] = false;
  if (off || !enable) return 0;
  GUARD(p);
  for (int w = 0; w < 2; ++w)
    if (int rc = DISPATCH(p, scan_material<float>(p, w), scan_material<double>(p, w))) return rc;
  return (p->mat_const[0] ? 1 : 0) | (p->mat_const[1] ? 2 : 0);
}

int fdtdhip_plan_set_step_buffers(fdtdhip_plan* p, void* E_alt, void* H_alt) {
  if (!p) return fail(-1, "null plan");
  p->E_alt = E_alt;
  p->H_alt = H_alt;
  return 0;
}

int fdtdhip_plan_set_pml_step_buffer(fdtdhip_plan* p, int id, void* psiE_alt) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->pmls.size()) return fail(-1, "unknown PML id");
  p->pmls[id].psiE_alt = psiE_alt;
  p->pml_dirty = true;
  return 0;
}

int fdtdhip_plan_set_halo(fdtdhip_plan* p, const void* H_below, const void* E_above) {
  if (!p) return fail(-1, "null plan");
  p->H_below = H_below;
  p->E_above = E_above;
  return 0;
}

int fdtdhip_plan_add_pml(fdtdhip_plan* p, int axis, int side, int t, const double* bE,
                         const double* cE, const double* bH, const double* cH, void* psiE,
                         void* psiH) {
  if (!p) return fail(-1, "null plan");
  if (axis < 0 || axis > 2 || (side != 0 && side != 1)) return fail(-1, "bad PML placement");
  const int n = axis == 0 ? p->nx : (axis == 1 ? p->ny : p->nz);
  if (t < 1 || t > n) return fail(-1, "PML thickness out of range");
  if (!bE || !cE || !bH || !cH || !psiE || !psiH) return fail(-1, "null PML array");
  if ((int)p->pmls.size() >= fdtdhip::MAX_PML) return fail(-1, "too many PMLs (one per face)");
  GUARD(p);
  PmlH h;
  h.axis = axis; h.side = side; h.t = t; h.lo = side == 0 ? 0 : n - t;
  h.psiE = psiE; h.psiH = psiH;
  h.fused = true;
  for (const BoundaryRef& b : p->boundaries)
    if (b.kind == 1) h.fused = false;
  int rc;
  if (p->dtype == FDTDHIP_F32) {
    if ((rc = upload_profile<float>(p, bE, t, &h.bE)) || (rc = upload_profile<float>(p, cE, t, &h.cE)) ||
        (rc = upload_profile<float>(p, bH, t, &h.bH)) || (rc = upload_profile<float>(p, cH, t, &h.cH)))
      return rc;
  } else {
    if ((rc = upload_profile<double>(p, bE, t, &h.bE)) || (rc = upload_profile<double>(p, cE, t, &h.cE)) ||
        (rc = upload_profile<double>(p, bH, t, &h.bH)) || (rc = upload_profile<double>(p, cH, t, &h.cH)))
      return rc;
  }
  p->pmls.push_back(h);
  p->pml_dirty = true;
  p->boundaries.push_back({0, (int)p->pmls.size() - 1});
  return (int)p->pmls.size() - 1;
}

int fdtdhip_plan_set_pml_state(fdtdhip_plan* p, int id, void* psiE, void* psiH) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->pmls.size()) return fail(-1, "unknown PML id");
  if (!psiE || !psiH) return fail(-1, "null PML array");
  PmlH& h = p->pmls[id];
  if (h.psiE == psiE && h.psiH == psiH) return 0;
  if (p->step_flipped) return fail(-1, "internal: PML state re-bound between the steps of a fused pair");
  h.psiE = psiE;
  h.psiH = psiH;
  p->pml_dirty = true;
  return 0;
}

int fdtdhip_plan_add_periodic(fdtdhip_plan* p, int axis) {
  if (!p) return fail(-1, "null plan");
  if (axis < 0 || axis > FDTDHIP_PERIODIC_X_EXTERNAL) return fail(-1, "bad axis");
  if (axis == FDTDHIP_PERIODIC_X_EXTERNAL)
    for (const BoundaryRef& b : p->boundaries)
      if (b.kind == 1 && b.idx == axis) return fail(-1, "one external periodic boundary at most");
  // Along an axis of one cell the first and the last plane are the same plane and the copy
  // (reference fdtd/boundaries.py:177-208) changes nothing: the usual z boundary of 2-D
  // grids.  Not recorded, so that such grids keep the one-launch step and the PMLs listed
  // after it keep their correction inside the bulk kernel.
  const int extent[3] = {p->nx, p->ny, p->nz};
  if (axis < 3 && extent[axis] == 1) return 0;
  p->boundaries.push_back({1, axis});
  return 0;
}

int fdtdhip_plan_add_object(fdtdhip_plan* p, int kind, const int32_t box[6], const void* inv,
                            const void* absorb) {
  if (!p) return fail(-1, "null plan");
  if (kind < 0 || kind > FDTDHIP_OBJ_ANISOTROPIC_MODULATED) return fail(-1, "unknown object kind");
  if (!box || !inv) return fail(-1, "null object array");
  if (kind == FDTDHIP_OBJ_ABSORBING && !absorb) return fail(-1, "absorbing object needs its factor");
  const int lim[3] = {p->nx, p->ny, p->nz};
  for (int d = 0; d < 3; ++d)
    if (box[2 * d] < 0 || box[2 * d + 1] > lim[d] || box[2 * d] > box[2 * d + 1])
      return fail(-1, "object box outside the grid");
  ObjH h;
  h.kind = kind;
  for (int i = 0; i < 6; ++i) h.box[i] = box[i];
  h.inv = inv;
  h.absorb = absorb;
  p->objs.push_back(h);
  p->obj_dirty = true;
  p->mat_const[0] = false;  // the box was zeroed after any earlier scan
  return (int)p->objs.size() - 1;
}

int fdtdhip_plan_add_source(fdtdhip_plan* p, int kind, int compE, int compH, int64_t n,
                            const int32_t* cells, const double* profile, double period,
                            double phase, int pulse, double frequency, double hanning_dt,
                            double cycle) {
  if (!p) return fail(-1, "null plan");
  if (!cells || !profile) return fail(-1, "null source array");
  if (compE < 0 || compE > 2 || compH < 0 || compH > 2) return fail(-1, "bad component");
  GUARD(p);
  SourceH s;
  s.kind = kind; s.compE = compE; s.compH = compH; s.n = n;
  s.period = period; s.phase = phase; s.pulse = pulse; s.frequency = frequency;
  s.hanning_dt = hanning_dt; s.cycle = cycle;
  const int lim[3] = {p->nx, p->ny, p->nz};
  if (kind == FDTDHIP_SRC_SET_BOX) {
    for (int d = 0; d < 3; ++d)
      if (cells[2 * d] < 0 || cells[2 * d + 1] > lim[d] || cells[2 * d] > cells[2 * d + 1])
        return fail(-1, "source box outside the grid");
    for (int i = 0; i < 6; ++i) s.box[i] = cells[i];
    s.amplitude = profile[0];
  } else if (kind == FDTDHIP_SRC_ADD_POINTS) {
    if (n < 0) return fail(-1, "negative point count");
    std::vector<int64_t> elem((size_t)n);
    std::set<int64_t> seen;
    for (int64_t i = 0; i < n; ++i) {
      for (int d = 0; d < 3; ++d)
        if (cells[3 * i + d] < 0 || cells[3 * i + d] >= lim[d])
          return fail(-1, "source point outside the grid");
      elem[i] = ((((int64_t)cells[3 * i] * p->ny) + cells[3 * i + 1]) * p->nz + cells[3 * i + 2]) * 3 + compE;
      if (!seen.insert(elem[i]).second) s.serial = 1;
      if (i == 0 || elem[i] < s.elem_min) s.elem_min = elem[i];
      if (i == 0 || elem[i] > s.elem_max) s.elem_max = elem[i];
    }
    s.elem_host = elem;
    s.profile_host.assign(profile, profile + n);
    if (n > 0) {
      HIPCHK(hipMalloc((void**)&s.elem, sizeof(int64_t) * (size_t)n));
      p->owned.push_back(s.elem);
      HIPCHK(hipMalloc((void**)&s.profile, sizeof(double) * (size_t)n));
      p->owned.push_back(s.profile);
      HIPCHK(hipMemcpy(s.elem, elem.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(s.profile, profile, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    }
  } else {
    return fail(-1, "unknown source kind");
  }
  p->sources.push_back(s);
  return 0;
}

int fdtdhip_plan_add_soft_source(fdtdhip_plan* p, const int32_t cell[3], const double* waveform,
                                 int64_t nwave, double impedance, double spacing) {
  if (!p) return fail(-1, "null plan");
  if (!cell || nwave < 0 || (nwave > 0 && !waveform)) return fail(-1, "null source array");
  if (spacing <= 0) return fail(-1, "bad spacing");
  const int lim[3] = {p->nx, p->ny, p->nz};
  for (int a = 0; a < 3; ++a)
    if (cell[a] < 0 || cell[a] >= lim[a]) return fail(-1, "source cell outside the grid");
  GUARD(p);
  SourceH s;
  s.kind = FDTDHIP_SRC_SOFT_POINT;
  s.compE = s.compH = 2; s.n = 1;
  s.period = 1; s.phase = 0; s.pulse = 0; s.frequency = 0; s.hanning_dt = 0; s.cycle = 1;
  for (int i = 0; i < 6; ++i) s.box[i] = 0;
  s.elem0 = ((((int64_t)cell[0] * p->ny) + cell[1]) * p->nz + cell[2]) * 3 + 2;  // Ez
  s.nwave = nwave; s.impedance = impedance; s.spacing = spacing;
  if (nwave > 0) {
    HIPCHK(hipMalloc((void**)&s.wave, sizeof(double) * (size_t)nwave));
    p->owned.push_back(s.wave);
    HIPCHK(hipMemcpy(s.wave, waveform, sizeof(double) * (size_t)nwave, hipMemcpyHostToDevice));
  }
  p->sources.push_back(s);
  return (int)p->sources.size() - 1;
}

int fdtdhip_plan_link_soft_source(fdtdhip_plan* p, int id, int detector_id) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->sources.size() || p->sources[id].kind != FDTDHIP_SRC_SOFT_POINT)
    return fail(-1, "not a soft source");
  if (detector_id < 0 || detector_id >= (int)p->dets.size() ||
      p->dets[detector_id].kind != FDTDHIP_DET_CURRENT)
    return fail(-1, "not a current detector");
  p->sources[id].det = detector_id;
  return 0;
}

int fdtdhip_plan_set_soft_source_io(fdtdhip_plan* p, int id, const void* prev_current,
                                    void* vin_out, void* vout_out, int64_t capacity) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->sources.size() || p->sources[id].kind != FDTDHIP_SRC_SOFT_POINT)
    return fail(-1, "not a soft source");
  SourceH& s = p->sources[id];
  s.prev_current = prev_current;
  s.vin = (char*)vin_out; s.vout = (char*)vout_out;
  s.vcap = capacity; s.vused = 0;
  return 0;
}

int fdtdhip_plan_set_soft_source_adjoint(fdtdhip_plan* p, int id, int64_t q0) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->sources.size() || p->sources[id].kind != FDTDHIP_SRC_SOFT_POINT)
    return fail(-1, "not a soft source");
  p->sources[id].adj_q0 = q0;
  return 0;
}

int fdtdhip_plan_add_modulated_source(fdtdhip_plan* p, const int32_t box[6], int comp,
                                      int channels) {
  if (!p) return fail(-1, "null plan");
  if (!box) return fail(-1, "null source array");
  if (comp < 0 || comp > 2) return fail(-1, "bad component");
  if (channels < 1) return fail(-1, "a modulated source needs at least one channel");
  const int lim[3] = {p->nx, p->ny, p->nz};
  for (int a = 0; a < 3; ++a)
    if (box[2 * a] < 0 || box[2 * a + 1] > lim[a] || box[2 * a] > box[2 * a + 1])
      return fail(-1, "source box outside the grid");
  SourceH s;
  s.kind = FDTDHIP_SRC_MODULATED_BOX;
  s.compE = comp; s.compH = comp; s.n = 1;
  s.period = 1; s.phase = 0; s.pulse = 0; s.frequency = 0; s.hanning_dt = 0; s.cycle = 1;
  for (int i = 0; i < 6; ++i) s.box[i] = box[i];
  s.channels = channels;
  p->sources.push_back(s);
  return (int)p->sources.size() - 1;
}

int fdtdhip_plan_set_modulated_source(fdtdhip_plan* p, int id, const void* P, const void* osc,
                                      int64_t osc_rows, int64_t q0, const void* weights,
                                      void* energy_trace, void* grad_P, void* grad_osc,
                                      void* grad_weights) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->sources.size() || p->sources[id].kind != FDTDHIP_SRC_MODULATED_BOX)
    return fail(-1, "not a modulated source");
  if (P && (!osc || !weights || osc_rows < 1)) return fail(-1, "modulated source: incomplete inputs");
  SourceH& s = p->sources[id];
  s.modP = P; s.osc = osc; s.osc_rows = osc_rows; s.q0 = q0; s.modw = weights;
  s.etrace = energy_trace; s.gP = grad_P; s.gosc = grad_osc; s.gwb = grad_weights;
  return 0;
}

int fdtdhip_plan_add_detector(fdtdhip_plan* p, int kind, int64_t n, const int32_t* cells,
                              double spacing) {
  if (!p) return fail(-1, "null plan");
  if (n < 0 || (n > 0 && !cells)) return fail(-1, "bad detector point list");
  GUARD(p);
  DetH d;
  d.kind = kind; d.n = n; d.spacing = spacing;
  const int lim[3] = {p->nx, p->ny, p->nz};
  for (int64_t i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c)
      if (cells[3 * i + c] < 0 || cells[3 * i + c] >= lim[c])
        return fail(-1, "detector point outside the grid");
  if (kind == FDTDHIP_DET_CELLS) {
    std::vector<int64_t> cell((size_t)n);
    for (int64_t i = 0; i < n; ++i)
      cell[i] = (((int64_t)cells[3 * i] * p->ny) + cells[3 * i + 1]) * p->nz + cells[3 * i + 2];
    d.cell_host = cell;
    if (n > 0) {
      HIPCHK(hipMalloc((void**)&d.cell, sizeof(int64_t) * (size_t)n));
      p->owned.push_back(d.cell);
      HIPCHK(hipMemcpy(d.cell, cell.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice));
    }
  } else if (kind == FDTDHIP_DET_CURRENT) {
    if (n > 0) {
      HIPCHK(hipMalloc((void**)&d.xyz, sizeof(int32_t) * 3 * (size_t)n));
      p->owned.push_back(d.xyz);
      HIPCHK(hipMemcpy(d.xyz, cells, sizeof(int32_t) * 3 * (size_t)n, hipMemcpyHostToDevice));
    }
  } else {
    return fail(-1, "unknown detector kind");
  }
  p->dets.push_back(d);
  p->detmap_state = 0;   // the fused step's cell -> detector map no longer covers every detector
  return (int)p->dets.size() - 1;
}

int fdtdhip_plan_set_detector_output(fdtdhip_plan* p, int id, void* outE, void* outH,
                                     int64_t capacity) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->dets.size()) return fail(-1, "unknown detector id");
  DetH& d = p->dets[id];
  d.outE = (char*)outE; d.outH = (char*)outH;
  d.capacity = capacity;
  d.usedE = d.usedH = 0;
  return 0;
}

int fdtdhip_plan_set_detector_below(fdtdhip_plan* p, int id, const void* H_plane) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->dets.size()) return fail(-1, "unknown detector id");
  if (p->dets[id].kind != FDTDHIP_DET_CURRENT) return fail(-1, "not a current detector");
  p->dets[id].below = H_plane;
  return 0;
}

int fdtdhip_read_deferred_currents(fdtdhip_plan* p) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  return DISPATCH(p, read_deferred_currents<float>(p), read_deferred_currents<double>(p));
}

int fdtdhip_update_E_bulk(fdtdhip_plan* p, int x0, int x1) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  return DISPATCH(p, bulk<float>(p, true, x0, x1), bulk<double>(p, true, x0, x1));
}
int fdtdhip_update_H_bulk(fdtdhip_plan* p, int x0, int x1) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  return DISPATCH(p, bulk<float>(p, false, x0, x1), bulk<double>(p, false, x0, x1));
}
int fdtdhip_update_E_post(fdtdhip_plan* p, int64_t q) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  return DISPATCH(p, post<float>(p, true, q), post<double>(p, true, q));
}
int fdtdhip_update_H_post(fdtdhip_plan* p, int64_t q) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  return DISPATCH(p, post<float>(p, false, q), post<double>(p, false, q));
}
int fdtdhip_update_E_post_stage(fdtdhip_plan* p, int64_t q, int stage) {
  if (!p) return fail(-1, "null plan");
  if (stage != 0 && stage != 1) return fail(-1, "stage is 0 or 1");
  GUARD(p);
  return DISPATCH(p, post<float>(p, true, q, POST_ALL, 0, -1, stage),
                  post<double>(p, true, q, POST_ALL, 0, -1, stage));
}
int fdtdhip_update_H_post_stage(fdtdhip_plan* p, int64_t q, int stage) {
  if (!p) return fail(-1, "null plan");
  if (stage != 0 && stage != 1) return fail(-1, "stage is 0 or 1");
  GUARD(p);
  return DISPATCH(p, post<float>(p, false, q, POST_ALL, 0, -1, stage),
                  post<double>(p, false, q, POST_ALL, 0, -1, stage));
}

int fdtdhip_update_E(fdtdhip_plan* p, int64_t q) {
  if (!p) return fail(-1, "null plan");
  if (int rc = fdtdhip_update_E_bulk(p, 0, p->nx)) return rc;
  return fdtdhip_update_E_post(p, q);
}
int fdtdhip_update_H(fdtdhip_plan* p, int64_t q) {
  if (!p) return fail(-1, "null plan");
  if (int rc = fdtdhip_update_H_bulk(p, 0, p->nx)) return rc;
  return fdtdhip_update_H_post(p, q);
}

int fdtdhip_run(fdtdhip_plan* p, int64_t q0, int64_t nsteps) {
  if (!p) return fail(-1, "null plan");
  if (p->stage_next != 0) return fail(-1, kStaged);
  int64_t s = 0;
  if (const FusedKind kind = fused_kind(p)) {
    // pairs of fused steps: first -> second buffers and back, so that the state ends
    // where the caller's tensors are; an odd last step runs as two half steps
    GUARD(p);
    if (p->poisoned) return fail(-5, kPoisoned);
    for (; s + 1 < nsteps; s += 2) {
      if (int rc = capacity_check(p, 2)) return rc;
      for (int k = 0; k < 2; ++k) {
        const int rc = DISPATCH(p, fused_step<float>(p, q0 + s + k, kind),
                                fused_step<double>(p, q0 + s + k, kind));
        if (rc) {
          // the state of the fields now sits in the second buffers (or between the two sets)
          if (p->step_flipped) p->poisoned = true;
          return rc;
        }
      }
    }
  }
  for (; s < nsteps; ++s) {
    if (int rc = fdtdhip_update_E(p, q0 + s)) return rc;
    if (int rc = fdtdhip_update_H(p, q0 + s)) return rc;
  }
  return 0;
}

int fdtdhip_plan_whole_step_kind(fdtdhip_plan* p, int staged, int32_t core[6], int32_t hole[6]) {
  if (!p) return fail(-1, "null plan");
  const FusedKind kind = fused_kind(p, staged != 0);
  int c[6] = {0, p->nx, 0, p->ny, 0, p->nz}, h[6] = {0, 0, 0, 0, 0, 0};
  const bool shell = !(p->pmls.empty() && p->objs.empty()) || p->H_below || p->E_above;
  if (kind == FUSED_MARCH && shell && !march_core(p, c, h)) return fail(-1, "internal: no core box");
  for (int n = 0; n < 6; ++n) {
    if (core) core[n] = c[n];
    if (hole) hole[n] = h[n];
  }
  return (int)kind;
}

int fdtdhip_step_stage(fdtdhip_plan* p, int64_t q, int stage) {
  if (!p) return fail(-1, "null plan");
  if (stage < 0 || stage > 3) return fail(-1, "stage is 0..3");
  GUARD(p);
  if (p->poisoned) return fail(-5, kPoisoned);
  if (stage != p->stage_next)
    return fail(-1, "fdtdhip_step_stage: stage " + std::to_string(p->stage_next) + " is due, not stage " +
                        std::to_string(stage));
  if (stage == 0) {
    if (!p->E || !p->H || !p->inv_eps || !p->inv_mu) return fail(-1, "fields not set");
    if (fused_kind(p, true) != FUSED_MARCH)
      return fail(-1, "this plan cannot take whole steps in stages (fdtdhip_plan_whole_step_kind)");
    if (int rc = capacity_check(p, 1)) return rc;
  }
  const int rc = DISPATCH(p, staged_step<float>(p, q, stage), staged_step<double>(p, q, stage));
  // (memory variables are advanced in place: a step that stops half way cannot be repeated)
  if (rc) p->poisoned = true;
  else p->stage_next = (stage + 1) % 4;
  return rc;
}

int fdtdhip_plan_set_object_trace(fdtdhip_plan* p, int id, void* trace, int64_t capacity) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->objs.size()) return fail(-1, "unknown object id");
  if (p->objs[id].kind != FDTDHIP_OBJ_ANISOTROPIC &&
      p->objs[id].kind != FDTDHIP_OBJ_ANISOTROPIC_MODULATED)
    return fail(-1, "curl traces exist for anisotropic objects only");
  p->objs[id].trace = trace;
  p->objs[id].trace_capacity = trace ? capacity : 0;
  p->obj_dirty = true;
  p->trace_cursor = -1;
  for (const ObjH& o : p->objs)
    if (o.trace && o.trace_capacity > 0) p->trace_cursor = 0;
  return 0;
}

int fdtdhip_plan_set_object_modifier(fdtdhip_plan* p, int id, const void* weights,
                                     void* energy_trace, void* grad_weights) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->objs.size()) return fail(-1, "unknown object id");
  if (p->objs[id].kind != FDTDHIP_OBJ_ANISOTROPIC_MODULATED)
    return fail(-1, "not a modulated anisotropic object");
  if (!weights) return fail(-1, "null modifier weights");
  p->objs[id].modw = weights;
  p->objs[id].etrace = energy_trace;
  p->objs[id].gmod = grad_weights;
  p->obj_dirty = true;
  return 0;
}

int fdtdhip_plan_set_object_grad(fdtdhip_plan* p, int id, void* grad) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->objs.size()) return fail(-1, "unknown object id");
  p->objs[id].grad = grad;
  p->obj_dirty = true;
  return 0;
}

int fdtdhip_plan_set_pml_adjoint(fdtdhip_plan* p, int id, void* lam_psiE, void* lam_psiH) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->pmls.size()) return fail(-1, "unknown PML id");
  p->pmls[id].lamE = lam_psiE;
  p->pmls[id].lamH = lam_psiH;
  p->pml_dirty = true;
  return 0;
}

int fdtdhip_adjoint_run(fdtdhip_plan* p, int64_t nsteps, void* lamE, void* lamH, void* work) {
  if (!p) return fail(-1, "null plan");
  if (!lamE || !lamH || !work) return fail(-1, "null adjoint array");
  GUARD(p);
  return DISPATCH(p, adjoint_run<float>(p, nsteps, lamE, lamH, work),
                  adjoint_run<double>(p, nsteps, lamE, lamH, work));
}

int fdtdhip_adjoint_run_noalloc(fdtdhip_plan* p, int64_t nsteps, void* lamE, void* lamH) {
  if (!p) return fail(-1, "null plan");
  if (!lamE || !lamH) return fail(-1, "null adjoint array");
  GUARD(p);
  return DISPATCH(p, adjoint_run_noalloc<float>(p, nsteps, lamE, lamH),
                  adjoint_run_noalloc<double>(p, nsteps, lamE, lamH));
}

int fdtdhip_adjoint_cells(fdtdhip_plan* p, int phase_E, int64_t step, void* lamE, void* lamH,
                          void* work) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  p->trace_cursor = step;
  const int rc = DISPATCH(p, adjoint_cells<float>(p, phase_E != 0, (float*)lamE, (float*)lamH, (float*)work),
                          adjoint_cells<double>(p, phase_E != 0, (double*)lamE, (double*)lamH, (double*)work));
  p->trace_cursor = -1;
  return rc;
}

int fdtdhip_adjoint_post_stage(fdtdhip_plan* p, int phase_E, int64_t step, int stage, void* lamE,
                               void* lamH) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  if (stage != 0 && stage != 1) return fail(-1, "stage must be 0 or 1");
  p->trace_cursor = step;
  const int rc = DISPATCH(p, adjoint_post<float>(p, phase_E != 0, (float*)lamE, (float*)lamH, stage),
                          adjoint_post<double>(p, phase_E != 0, (double*)lamE, (double*)lamH, stage));
  p->trace_cursor = -1;
  return rc;
}

int fdtdhip_adjoint_gather(fdtdhip_plan* p, int phase_E, void* lamE, void* lamH, void* work,
                           const void* work_ghost) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  return DISPATCH(p, adjoint_gather<float>(p, phase_E != 0, (float*)lamE, (float*)lamH, (float*)work,
                                           (const float*)work_ghost),
                  adjoint_gather<double>(p, phase_E != 0, (double*)lamE, (double*)lamH, (double*)work,
                                         (const double*)work_ghost));
}

int fdtdhip_plane_energy(fdtdhip_plan* p, int axis, int index, void* outE, void* outH) {
  if (!p) return fail(-1, "null plan");
  if (!p->E || !p->H) return fail(-1, "fields not set");
  if (!outE || !outH) return fail(-1, "null output");
  const int n = axis == 0 ? p->nx : (axis == 1 ? p->ny : p->nz);
  if (axis < 0 || axis > 2 || index < 0 || index >= n) return fail(-1, "plane outside the grid");
  GUARD(p);
  const int64_t cells = p->cells() / n;
  if (p->dtype == FDTDHIP_F32)
    hipLaunchKernelGGL((plane_energy<float>), dim3(blocks_for(cells)), dim3(256), 0, p->stream,
                       (const float*)p->E, (const float*)p->H, p->nx, p->ny, p->nz, axis, index,
                       (float*)outE, (float*)outH);
  else
    hipLaunchKernelGGL((plane_energy<double>), dim3(blocks_for(cells)), dim3(256), 0, p->stream,
                       (const double*)p->E, (const double*)p->H, p->nx, p->ny, p->nz, axis, index,
                       (double*)outE, (double*)outH);
  HIPCHK(hipGetLastError());
  return 0;
}

int fdtdhip_plan_enable_timing(fdtdhip_plan* p, int on) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  if (int rc = drain_timing(p)) return rc;
  p->timing = on != 0;
  for (int n = 0; n < 4; ++n) { p->ms[n] = 0; p->launches[n] = 0; }
  return 0;
}

int fdtdhip_plan_get_timing_parts(fdtdhip_plan* p, double ms[4], int64_t launches[4]) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  if (int rc = drain_timing(p)) return rc;
  for (int n = 0; n < 4; ++n) {
    if (ms) ms[n] = p->ms[n];
    if (launches) launches[n] = p->launches[n];
    p->ms[n] = 0;
    p->launches[n] = 0;
  }
  return 0;
}

int fdtdhip_plan_get_timing(fdtdhip_plan* p, double* ms_E, int64_t* launches_E, double* ms_H,
                            int64_t* launches_H) {
  double ms[4];
  int64_t n[4];
  if (int rc = fdtdhip_plan_get_timing_parts(p, ms, n)) return rc;
  // whole-step kernels and their shell launches are reported with the E half step
  if (ms_E) *ms_E = ms[TIME_E] + ms[TIME_WHOLE] + ms[TIME_SHELL];
  if (launches_E) *launches_E = n[TIME_E] + n[TIME_WHOLE];
  if (ms_H) *ms_H = ms[TIME_H];
  if (launches_H) *launches_H = n[TIME_H];
  return 0;
}

int fdtdhip_curl_E(const void* E, void* out, int nx, int ny, int nz, int dtype, void* stream) {
  if (!E || !out) return fail(-1, "null pointer");
  return dtype == FDTDHIP_F32 ? curl_launch<float>(E, out, nx, ny, nz, true, stream)
                              : curl_launch<double>(E, out, nx, ny, nz, true, stream);
}
int fdtdhip_curl_H(const void* H, void* out, int nx, int ny, int nz, int dtype, void* stream) {
  if (!H || !out) return fail(-1, "null pointer");
  return dtype == FDTDHIP_F32 ? curl_launch<float>(H, out, nx, ny, nz, false, stream)
                              : curl_launch<double>(H, out, nx, ny, nz, false, stream);
}

}  // extern "C"
