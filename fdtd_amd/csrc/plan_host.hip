// Host-side plan state of libfdtd_hip that launches no kernels: error reporting, release of
// the plan's tables, stream-ordered table uploads, kernel-strategy knobs, event timing, source
// waveforms (reference fdtd/sources.py:99-110).
#include "plan_host.hpp"

namespace fdtdhip {

namespace {
thread_local std::string g_err;
}

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

const char* last_error() { return g_err.c_str(); }

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
  p->ovr_obj = -1;
  p->mat_const[0] = p->mat_const[1] = false;
  p->detmap = nullptr;  // was in `owned`
  p->detmap_state = 0;
  p->adj_work = nullptr;  // was in `owned`
  p->adj_work_bytes = 0;
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

// The E pass of a step filled one row of the objects' curl traces.
int64_t two_step_launches = 0;


int64_t trace_rows_left(const fdtdhip_plan* p) {
  int64_t left = INT64_MAX;
  if (p->trace_cursor < 0) return left;
  for (const ObjH& o : p->objs)
    if (o.trace) left = std::min<int64_t>(left, o.trace_capacity - p->trace_cursor);
  return left;
}

void advance_trace_cursor(fdtdhip_plan* p) {
  if (p->trace_cursor < 0) return;
  ++p->trace_cursor;
  for (const ObjH& o : p->objs)
    if (o.trace && p->trace_cursor >= o.trace_capacity) p->trace_cursor = -1;  // traces full
}

// Index of the periodic boundary along x whose two planes live on different ranks, or the
// number of boundaries.
size_t external_periodic(const fdtdhip_plan* p) {
  for (size_t b = 0; b < p->boundaries.size(); ++b)
    if (p->boundaries[b].kind == 1 && p->boundaries[b].idx == FDTDHIP_PERIODIC_X_EXTERNAL) return b;
  return p->boundaries.size();
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

}  // namespace fdtdhip
