// libfdtd_hip: C ABI (include/fdtd_hip.h) over the HIP kernels of this directory.
//
// A plan keeps the borrowed field pointers, the ordered boundary list, the object / source /
// detector tables (plan_host.hpp), and turns one half step of the reference's Grid (reference
// fdtd/grid.py:275-325) into a short, fixed sequence of kernel launches on the plan's stream
// (launch_half.hip, launch_whole.hip, launch_adjoint.hip).
#include "plan_host.hpp"

using namespace fdtdhip;

extern "C" {

int fdtdhip_abi_version(void) { return FDTDHIP_ABI_VERSION; }

const char* fdtdhip_last_error(void) { return last_error(); }

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
    if (int rc = half_scan_material(p, w)) return rc;
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

int fdtdhip_plan_bind(fdtdhip_plan* p, void* stream, void* E, void* H, const void* inv_eps,
                      const void* inv_mu, void* E_alt, void* H_alt, const void* H_below,
                      const void* E_above) {
  if (int rc = fdtdhip_plan_set_fields(p, E, H, inv_eps, inv_mu)) return rc;
  p->stream = (hipStream_t)stream;
  p->E_alt = E_alt;
  p->H_alt = H_alt;
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
    for (int i = 0; i < 6; ++i) s.box[i] = 0;      // the box that holds the points
    for (int64_t i = 0; i < n; ++i) {
      for (int d = 0; d < 3; ++d) {
        if (cells[3 * i + d] < 0 || cells[3 * i + d] >= lim[d])
          return fail(-1, "source point outside the grid");
        s.box[2 * d] = i == 0 ? cells[3 * i + d] : std::min(s.box[2 * d], cells[3 * i + d]);
        s.box[2 * d + 1] = i == 0 ? cells[3 * i + d] + 1 : std::max(s.box[2 * d + 1], cells[3 * i + d] + 1);
      }
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

int fdtdhip_plan_set_soft_source_adjoint_below(fdtdhip_plan* p, int id, void* lamH_below) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->sources.size() || p->sources[id].kind != FDTDHIP_SRC_SOFT_POINT)
    return fail(-1, "not a soft source");
  p->sources[id].adj_below = lamH_below;
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
  return half_read_deferred_currents(p);
}

int fdtdhip_update_E_bulk(fdtdhip_plan* p, int x0, int x1) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  return half_bulk(p, true, x0, x1);
}
int fdtdhip_update_H_bulk(fdtdhip_plan* p, int x0, int x1) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  return half_bulk(p, false, x0, x1);
}
int fdtdhip_update_E_post(fdtdhip_plan* p, int64_t q) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  return half_post(p, true, q);
}
int fdtdhip_update_H_post(fdtdhip_plan* p, int64_t q) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  return half_post(p, false, q);
}
int fdtdhip_update_E_post_stage(fdtdhip_plan* p, int64_t q, int stage) {
  if (!p) return fail(-1, "null plan");
  if (stage != 0 && stage != 1) return fail(-1, "stage is 0 or 1");
  GUARD(p);
  return half_post(p, true, q, POST_ALL, 0, -1, stage);
}
int fdtdhip_update_H_post_stage(fdtdhip_plan* p, int64_t q, int stage) {
  if (!p) return fail(-1, "null plan");
  if (stage != 0 && stage != 1) return fail(-1, "stage is 0 or 1");
  GUARD(p);
  return half_post(p, false, q, POST_ALL, 0, -1, stage);
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
  const FusedKind kind = fused_kind(p);
  const bool pairs = nsteps >= 4 && whole_pair_possible(p);
  if (kind || pairs) {
    // pairs of fused steps: first -> second buffers and back, so that the state ends
    // where the caller's tensors are; an odd last step runs as two half steps
    GUARD(p);
    if (p->poisoned) return fail(-5, kPoisoned);
    // plans with nothing between their steps but source points: two steps per pass, in
    // pairs of passes (first -> second buffers and back)
    if (pairs)
      for (; s + 3 < nsteps && trace_rows_left(p) >= 4; s += 4) {
        if (int rc = capacity_check(p, 4)) return rc;      // (before anything is enqueued)
        for (int k = 0; k < 2; ++k)
          if (const int rc = whole_pair_step(p, q0 + s + 2 * k)) {
            // (the cells of PML faces and of the objects' box take the first step of a pair in
            // place, memory variables included: a pass that stops half way cannot be repeated)
            if (p->step_flipped || !p->pmls.empty() || !p->objs.empty()) p->poisoned = true;
            return rc;
          }
      }
    for (; kind && s + 1 < nsteps; s += 2) {
      if (int rc = capacity_check(p, 2)) return rc;
      for (int k = 0; k < 2; ++k) {
        const int rc = whole_fused_step(p, q0 + s + k, kind);
        if (rc) {
          // the state of the fields now sits in the second buffers (or between the two sets);
          // the marching step advances PML memory variables in place as it goes, so with
          // faces a failed step cannot be repeated whichever buffers hold the fields
          if (p->step_flipped || (kind == FUSED_MARCH && !p->pmls.empty())) p->poisoned = true;
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

int fdtdhip_step_into(fdtdhip_plan* p, int64_t q) {
  if (!p) return fail(-1, "null plan");
  if (p->stage_next != 0) return fail(-1, kStaged);
  GUARD(p);
  if (p->poisoned) return fail(-5, kPoisoned);
  if (p->step_flipped) return fail(-1, "internal: fdtdhip_step_into between the steps of a fused pair");
  const FusedKind kind = fused_kind(p);
  if (!kind) return fail(-1, "this plan cannot take a whole step in one pass (fdtdhip_plan_whole_step_kind)");
  if (int rc = capacity_check(p, 1)) return rc;
  if (const int rc = whole_fused_step(p, q, kind)) {
    if (kind == FUSED_MARCH && !p->pmls.empty()) p->poisoned = true;   // memory variables advanced in place
    else if (p->step_flipped) { std::swap(p->E, p->E_alt); std::swap(p->H, p->H_alt); p->step_flipped = false; }
    return rc;
  }
  // the roles are exchanged: the step buffers hold the state, the former fields are the step
  // buffers; the one-lane-per-cell kernel wrote psi_E of every PML into its step buffer too
  // (the marching kernel's shell advances it in place)
  if (kind != FUSED_MARCH)
    for (PmlH& h : p->pmls) std::swap(h.psiE, h.psiE_alt);
  p->pml_dirty = true;
  p->step_flipped = false;
  return (int)kind;
}

int fdtdhip_plan_whole_step_kind(fdtdhip_plan* p, int staged, int32_t core[6], int32_t hole[6]) {
  if (!p) return fail(-1, "null plan");
  const FusedKind kind = fused_kind(p, staged != 0);
  int c[6] = {0, p->nx, 0, p->ny, 0, p->nz}, h[6] = {0, 0, 0, 0, 0, 0};
  const bool shell = !(p->pmls.empty() && p->objs.empty()) || p->H_below || p->E_above || sources_by_launch(p) ||
                     periodic_by_launch(p);
  if (kind == FUSED_MARCH && shell && !march_core(p, c, h)) return fail(-1, "internal: no core box");
  for (int n = 0; n < 6; ++n) {
    if (core) core[n] = c[n];
    if (hole) hole[n] = h[n];
  }
  return (int)kind;
}

int64_t fdtdhip_two_step_launches(void) { return fdtdhip::two_step_launches; }

int fdtdhip_plan_two_step_box(fdtdhip_plan* p, int staged, int32_t inner[6]) {
  if (!p) return fail(-1, "null plan");
  return whole_pair_possible(p, inner, staged != 0) ? 1 : 0;
}

int fdtdhip_pair_stage(fdtdhip_plan* p, int64_t q, int stage) {
  if (!p) return fail(-1, "null plan");
  if (stage < 0 || stage > 7) return fail(-1, "stage is 0..7");
  GUARD(p);
  if (p->poisoned) return fail(-5, kPoisoned);
  // (p->stage_next: 0 = no step in progress, 1..3 = fdtdhip_step_stage, 11..17 = this call)
  const int due = stage == 0 ? 0 : 10 + stage;
  if (p->stage_next != due)
    return fail(-1, "fdtdhip_pair_stage: stage " + std::to_string(stage) + " is not the piece that is due");
  if (stage == 0) {
    if (!p->E || !p->H || !p->inv_eps || !p->inv_mu) return fail(-1, "fields not set");
    if (!whole_pair_possible(p, nullptr, true))
      return fail(-1, "this plan cannot take two steps per pass in stages (fdtdhip_plan_two_step_box)");
    if (int rc = capacity_check(p, 2)) return rc;      // (nothing enqueued yet: the plan stays usable)
  }
  const int rc = whole_pair_stage(p, q, stage);
  // (fields and memory variables are advanced in place: a pair that stops half way cannot be repeated)
  if (rc) p->poisoned = true;
  else p->stage_next = stage == 7 ? 0 : 10 + stage + 1;
  return rc;
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
  const int rc = whole_staged_step(p, q, stage);
  // (memory variables are advanced in place: a step that stops half way cannot be repeated)
  if (rc) p->poisoned = true;
  else p->stage_next = (stage + 1) % 4;
  return rc;
}

namespace {
// The trace / gradient arrays of object `id` changed.  While the table on the device is
// otherwise current, one object's arrays go with the kernel arguments and the table stays as
// it is; a second object, or a table that has to be written anyway, takes the upload.
void rebind_object_arrays(fdtdhip_plan* p, int id) {
  if (!p->obj_dirty && (p->ovr_obj < 0 || p->ovr_obj == id)) p->ovr_obj = id;
  else p->obj_dirty = true;
}
}  // namespace

int fdtdhip_plan_set_object_trace(fdtdhip_plan* p, int id, void* trace, int64_t capacity) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->objs.size()) return fail(-1, "unknown object id");
  if (p->objs[id].kind != FDTDHIP_OBJ_ANISOTROPIC &&
      p->objs[id].kind != FDTDHIP_OBJ_ANISOTROPIC_MODULATED)
    return fail(-1, "curl traces exist for anisotropic objects only");
  p->objs[id].trace = trace;
  p->objs[id].trace_capacity = trace ? capacity : 0;
  rebind_object_arrays(p, id);
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
  rebind_object_arrays(p, id);
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

int fdtdhip_plan_set_adjoint_buffers(fdtdhip_plan* p, void* lamE_alt, void* lamH_alt) {
  if (!p) return fail(-1, "null plan");
  p->adjE_alt = lamE_alt;
  p->adjH_alt = lamH_alt;
  return 0;
}

int fdtdhip_plan_set_pml_adjoint_buffers(fdtdhip_plan* p, int id, void* lam_psiE_alt, void* lam_psiH_alt) {
  if (!p) return fail(-1, "null plan");
  if (id < 0 || id >= (int)p->pmls.size()) return fail(-1, "unknown PML id");
  p->pmls[id].lamE_alt = lam_psiE_alt;
  p->pmls[id].lamH_alt = lam_psiH_alt;
  p->pml_dirty = true;
  return 0;
}

int fdtdhip_adjoint_run(fdtdhip_plan* p, int64_t nsteps, void* lamE, void* lamH, void* work) {
  if (!p) return fail(-1, "null plan");
  if (!lamE || !lamH || !work) return fail(-1, "null adjoint array");
  GUARD(p);
  return adj_run(p, nsteps, lamE, lamH, work);
}

int fdtdhip_plan_adjoint_whole_steps(fdtdhip_plan* p) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  return adj_stage_possible(p);
}

int fdtdhip_adjoint_pack_plane(fdtdhip_plan* p, int phase_E, int64_t step, int flip, void* lamE, void* lamH,
                               void* out) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  p->trace_cursor = step;
  const int rc = adj_pack(p, phase_E != 0, lamE, lamH, out, flip);
  p->trace_cursor = -1;
  return rc;
}

int fdtdhip_adjoint_step_stage(fdtdhip_plan* p, int64_t step, int stage, int flip, void* lamE0, void* lamH0,
                               void* lamE1, void* lamH1, void* work, const void* ghost) {
  if (!p) return fail(-1, "null plan");
  if (stage < 0 || stage > 3) return fail(-1, "stage is 0..3");
  GUARD(p);
  p->trace_cursor = step;
  const int rc = adj_stage(p, stage, lamE0, lamH0, lamE1, lamH1, work, flip, ghost);
  p->trace_cursor = -1;
  return rc;
}

int fdtdhip_adjoint_run_noalloc(fdtdhip_plan* p, int64_t nsteps, void* lamE, void* lamH) {
  if (!p) return fail(-1, "null plan");
  if (!lamE || !lamH) return fail(-1, "null adjoint array");
  GUARD(p);
  return adj_run_noalloc(p, nsteps, lamE, lamH);
}

int fdtdhip_adjoint_cells(fdtdhip_plan* p, int phase_E, int64_t step, void* lamE, void* lamH,
                          void* work) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  p->trace_cursor = step;
  const int rc = adj_cells(p, phase_E != 0, lamE, lamH, work);
  p->trace_cursor = -1;
  return rc;
}

int fdtdhip_adjoint_post_stage(fdtdhip_plan* p, int phase_E, int64_t step, int stage, void* lamE,
                               void* lamH) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  if (stage != 0 && stage != 1) return fail(-1, "stage must be 0 or 1");
  p->trace_cursor = step;
  const int rc = adj_post(p, phase_E != 0, lamE, lamH, stage);
  p->trace_cursor = -1;
  return rc;
}

int fdtdhip_adjoint_gather(fdtdhip_plan* p, int phase_E, void* lamE, void* lamH, void* work,
                           const void* work_ghost) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  return adj_gather(p, phase_E != 0, lamE, lamH, work, work_ghost);
}

int fdtdhip_plane_energy(fdtdhip_plan* p, int axis, int index, void* outE, void* outH) {
  if (!p) return fail(-1, "null plan");
  if (!p->E || !p->H) return fail(-1, "fields not set");
  if (!outE || !outH) return fail(-1, "null output");
  const int n = axis == 0 ? p->nx : (axis == 1 ? p->ny : p->nz);
  if (axis < 0 || axis > 2 || index < 0 || index >= n) return fail(-1, "plane outside the grid");
  GUARD(p);
  return half_plane_energy(p, axis, index, outE, outH);
}

int fdtdhip_plan_enable_timing(fdtdhip_plan* p, int on) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  if (int rc = drain_timing(p)) return rc;
  p->timing = on != 0;
  for (int n = 0; n < TIME_KINDS; ++n) { p->ms[n] = 0; p->launches[n] = 0; }
  return 0;
}

int fdtdhip_plan_get_timing_parts(fdtdhip_plan* p, double ms[5], int64_t launches[5]) {
  if (!p) return fail(-1, "null plan");
  GUARD(p);
  if (int rc = drain_timing(p)) return rc;
  for (int n = 0; n < TIME_KINDS; ++n) {
    if (ms) ms[n] = p->ms[n];
    if (launches) launches[n] = p->launches[n];
    p->ms[n] = 0;
    p->launches[n] = 0;
  }
  return 0;
}

int fdtdhip_plan_get_timing(fdtdhip_plan* p, double* ms_E, int64_t* launches_E, double* ms_H,
                            int64_t* launches_H) {
  double ms[TIME_KINDS];
  int64_t n[TIME_KINDS];
  if (int rc = fdtdhip_plan_get_timing_parts(p, ms, n)) return rc;
  // whole-step kernels and their shell launches are reported with the E half step
  if (ms_E) *ms_E = ms[TIME_E] + ms[TIME_WHOLE] + ms[TIME_SHELL] + ms[TIME_PAIR];
  if (launches_E) *launches_E = n[TIME_E] + n[TIME_WHOLE] + n[TIME_PAIR];
  if (ms_H) *ms_H = ms[TIME_H];
  if (launches_H) *launches_H = n[TIME_H];
  return 0;
}

int fdtdhip_curl_E(const void* E, void* out, int nx, int ny, int nz, int dtype, void* stream) {
  if (!E || !out) return fail(-1, "null pointer");
  return half_curl(E, out, nx, ny, nz, dtype, true, stream);
}
int fdtdhip_curl_H(const void* H, void* out, int nx, int ny, int nz, int dtype, void* stream) {
  if (!H || !out) return fail(-1, "null pointer");
  return half_curl(H, out, nx, ny, nz, dtype, false, stream);
}

}  // extern "C"
