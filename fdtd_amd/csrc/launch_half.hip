// Half steps of libfdtd_hip (reference fdtd/grid.py:275-325): the fused bulk pass over a range
// of x planes, the post pieces that follow it (periodic copies, PMLs registered after a
// periodic boundary, sources, detectors), the material scan, stand-alone curls.
#include "plan_host.hpp"
#include "yee_generic.hpp"
#include "yee_tiled.hpp"

namespace fdtdhip {

namespace {

template <typename T>
int bulk_launch(fdtdhip_plan* p, bool phaseE, int x0, int x1);

template <typename T>
int bulk(fdtdhip_plan* p, bool phaseE, int x0, int x1) {
  if (!p->timing || x0 == x1) return bulk_launch<T>(p, phaseE, x0, x1);
  size_t slot;
  if (int rc = span_begin(p, slot)) return rc;
  const int rc = bulk_launch<T>(p, phaseE, x0, x1);
  if (int rc2 = span_end(p, slot, phaseE ? TIME_E : TIME_H)) return rc2;
  return rc;
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
  fill_object_override<T>(p, a);
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

}  // namespace

int half_bulk(fdtdhip_plan* p, bool phaseE, int x0, int x1) {
  return DISPATCH(p, bulk<float>(p, phaseE, x0, x1), bulk<double>(p, phaseE, x0, x1));
}

int half_post(fdtdhip_plan* p, bool phaseE, int64_t q, int parts, int x0, int x1, int stage) {
  return DISPATCH(p, post<float>(p, phaseE, q, parts, x0, x1, stage),
                  post<double>(p, phaseE, q, parts, x0, x1, stage));
}

namespace {
// Additive point lists and boxes that are set, at time index q, applied to F (post() applies
// the same launches to the plan's current arrays, among the other kinds of source).
template <typename T>
int sources_to(fdtdhip_plan* p, bool phaseE, int64_t q, T* F) {
  for (const SourceH& s : p->sources) {
    const double w = waveform(s, q);
    if (s.kind == FDTDHIP_SRC_ADD_POINTS) {
      if (!phaseE || s.n == 0) continue;
      hipLaunchKernelGGL((source_add_points<T>), dim3(s.serial ? 1 : blocks_for(s.n, 64)), dim3(64), 0,
                         p->stream, F, s.elem, s.profile, w, s.n, s.serial, (int64_t)0, p->cells() * 3);
    } else if (s.kind == FDTDHIP_SRC_SET_BOX) {
      const int64_t total = (int64_t)(s.box[1] - s.box[0]) * (s.box[3] - s.box[2]) * (s.box[5] - s.box[4]);
      if (total <= 0) continue;
      hipLaunchKernelGGL((source_set_box<T>), dim3(blocks_for(total)), dim3(256), 0, p->stream, F, p->ny,
                         p->nz, s.box[0], s.box[1], s.box[2], s.box[3], s.box[4], s.box[5],
                         phaseE ? s.compE : s.compH, (T)(s.amplitude * w));
    } else {
      return fail(-1, "internal: a source of this kind is not applied apart from its half step");
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}
}  // namespace

namespace {
template <typename T>
int periodic_to(fdtdhip_plan* p, bool phaseE, T* F) {
  for (const BoundaryRef& b : p->boundaries) {
    if (b.kind != 1 || b.idx > 2) continue;
    const int axis = b.idx;
    const int n = axis == 0 ? p->nx : (axis == 1 ? p->ny : p->nz);
    if (n < 2) continue;
    const int64_t total = p->cells() / n * 3;
    const int dst = phaseE ? 0 : n - 1, src = phaseE ? n - 1 : 0;
    hipLaunchKernelGGL((periodic_copy<T>), dim3(blocks_for(total)), dim3(256), 0, p->stream, F, p->nx, p->ny,
                       p->nz, axis, dst, src);
  }
  HIPCHK(hipGetLastError());
  return 0;
}
}  // namespace

int apply_periodic_to(fdtdhip_plan* p, bool phaseE, void* F) {
  return DISPATCH(p, periodic_to<float>(p, phaseE, (float*)F), periodic_to<double>(p, phaseE, (double*)F));
}

int apply_sources_to(fdtdhip_plan* p, bool phaseE, int64_t q, void* F) {
  return DISPATCH(p, sources_to<float>(p, phaseE, q, (float*)F), sources_to<double>(p, phaseE, q, (double*)F));
}

int half_read_deferred_currents(fdtdhip_plan* p) {
  return DISPATCH(p, read_deferred_currents<float>(p), read_deferred_currents<double>(p));
}

int half_scan_material(fdtdhip_plan* p, int w) {
  return DISPATCH(p, scan_material<float>(p, w), scan_material<double>(p, w));
}

int half_curl(const void* in, void* out, int nx, int ny, int nz, int dtype, bool forward, void* stream) {
  return dtype == FDTDHIP_F32 ? curl_launch<float>(in, out, nx, ny, nz, forward, stream)
                              : curl_launch<double>(in, out, nx, ny, nz, forward, stream);
}

int half_plane_energy(fdtdhip_plan* p, int axis, int index, void* outE, void* outH) {
  const int n = axis == 0 ? p->nx : (axis == 1 ? p->ny : p->nz);
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

}  // namespace fdtdhip
