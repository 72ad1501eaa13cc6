// Whole steps of libfdtd_hip: one launch (or a marching kernel with its shell launches) per
// time step instead of the reference's two passes (reference fdtd/grid.py:267-273), and the
// policy that decides which plans take them.
#include "plan_host.hpp"
#include "yee_generic.hpp"
#include "yee_step.hpp"
#include "yee_march.hpp"
#include "yee_march2.hpp"
#include "yee_shell.hpp"

namespace fdtdhip {

// ---- fused full step (yee_step.hpp) -------------------------------------------------
// Usable when nothing has to happen between the two half steps that the fused
// kernel cannot do itself: no periodic boundary, no x-slab halo, no curl trace being
// recorded, at most MAX_STEP_SOURCES additive source points without repeated cells, and the
// caller has provided the second buffers.  (The marching kernels below take more: open slab
// faces in stages, recorded runs, and -- on undivided grids -- sources of any number of
// points and plane sources that are set, applied by launches of their own: sources_by_launch.)
constexpr int64_t FUSED_AUTO_MAX_CELLS = 1 << 17;

#ifndef FDTDHIP_MARCH_TY32
#define FDTDHIP_MARCH_TY32 4   // rows per workgroup of the marching kernel in float32 (experiments: -D)
#endif
#ifndef FDTDHIP_MARCH_TY64
#define FDTDHIP_MARCH_TY64 4   // the same in float64
#endif
#ifndef FDTDHIP_MARCH2_TY32
#define FDTDHIP_MARCH2_TY32 4  // rows per wave of the two-step marching kernel in float32
#endif
// Rows per tile of the two-step kernels (MI355X, 512^3 vacuum, ms per launch of two steps).
// A workgroup per tile (z rows of up to 512 cells), float32: 2 / 3 / 4 / 5 / 6 / 7 / 8 rows
// 1.55 / 1.68 / 1.57 / 1.48 / 1.47 / 2.22 (spills) / 3.27; float64: 1 / 2 / 3 rows 4.96 / 3.23 /
// 5.80 (spills).  Independent waves, float32: 2 / 3 / 4 / 6 / 8 rows 1.82 / 1.68 / 1.63 / 1.64 /
// 2.13; float64: 2 / 3 rows 3.71 / 3.58, 4 spill.
#ifndef FDTDHIP_MARCH2_ROWS_TY32
#define FDTDHIP_MARCH2_ROWS_TY32 6   // rows per workgroup of the two-step kernel with a workgroup per tile
#endif
#ifndef FDTDHIP_MARCH2_ROWS_TY64
#define FDTDHIP_MARCH2_ROWS_TY64 2
#endif
#ifndef FDTDHIP_MARCH2_TY64
#define FDTDHIP_MARCH2_TY64 3  // the same in float64 (4: spills)
#endif

// Plans with PML faces take the marching step over the box no face reaches, with half steps
// of their own for the shell around it (yee_shell.hpp).  That needs a box.
// `hole`: the box that holds every object, clipped to the core (empty without objects): its E
// half step belongs to the shell as well.
namespace {
// Whether the marching kernel itself evaluates the E half step of the cell of element `elem`.
bool march_owns(const fdtdhip_plan* p, const int (&core)[6], const int (&hole)[6], int64_t elem) {
  const int64_t plane = (int64_t)p->ny * p->nz, cell = elem / 3;
  const int i = (int)(cell / plane), j = (int)((cell % plane) / p->nz), k = (int)(cell % p->nz);
  if (i < core[0] || i >= core[1] || j < core[2] || j >= core[3] || k < core[4] || k >= core[5])
    return false;
  return !(i >= hole[0] && i < hole[1] && j >= hole[2] && j < hole[3] && k >= hole[4] && k < hole[5]);
}

}  // namespace

bool sources_by_launch(const fdtdhip_plan* p) {
  int64_t points = 0;
  bool set_box = false;
  for (const SourceH& s : p->sources) {
    if (s.kind == FDTDHIP_SRC_SET_BOX) set_box = true;
    else if (s.kind == FDTDHIP_SRC_ADD_POINTS && !s.serial) points += s.n;
    else return false;
  }
  return set_box || points > MAX_STEP_SOURCES;
}

bool periodic_by_launch(const fdtdhip_plan* p) {
  for (const BoundaryRef& b : p->boundaries)
    if (b.kind == 1 && b.idx <= 2) return true;
  return false;
}

bool march_core(const fdtdhip_plan* p, int (&core)[6], int (&hole)[6]) {
  BulkArgs<float> b;
  fill_core<float>(p, b);
  for (int n = 0; n < 6; ++n) core[n] = b.core[n];
  // on an x-slab the plane that reads the neighbour's H (E half step of plane 0) or E (H half
  // step of the last plane) waits for the halo exchange: it belongs to the shell
  if (p->H_below) core[0] = std::max(core[0], 1);
  if (p->E_above) core[1] = std::min(core[1], p->nx - 1);
  // the two planes of a periodic boundary are copied onto each other after each half step:
  // they belong to the shell as well
  for (const BoundaryRef& pb : p->boundaries)
    if (pb.kind == 1 && pb.idx <= 2) {
      const int n = pb.idx == 0 ? p->nx : (pb.idx == 1 ? p->ny : p->nz);
      core[2 * pb.idx] = std::max(core[2 * pb.idx], 1);
      core[2 * pb.idx + 1] = std::min(core[2 * pb.idx + 1], n - 1);
    }
  // the box the shell launches step inside the core: the objects, and the cells of sources that
  // are applied by launches of their own between the half steps
  int hull[6];
  for (int n = 0; n < 6; ++n) hull[n] = b.hull[n];
  const bool apart = sources_by_launch(p);
  if (apart)
    for (const SourceH& s : p->sources) {
      if (s.box[1] <= s.box[0] || s.box[3] <= s.box[2] || s.box[5] <= s.box[4]) continue;
      for (int ax = 0; ax < 3; ++ax) {
        hull[2 * ax] = std::min(hull[2 * ax], s.box[2 * ax]);
        hull[2 * ax + 1] = std::max(hull[2 * ax + 1], s.box[2 * ax + 1]);
      }
    }
  for (int ax = 0; ax < 3; ++ax) {
    if (core[2 * ax + 1] - core[2 * ax] < 1) return false;
    hole[2 * ax] = std::min(std::max(hull[2 * ax], core[2 * ax]), core[2 * ax + 1]);
    hole[2 * ax + 1] = std::min(std::max(hull[2 * ax + 1], hole[2 * ax]), core[2 * ax + 1]);
  }
  if ((p->objs.empty() && !apart) || hole[1] == hole[0] || hole[3] == hole[2] || hole[5] == hole[4])
    for (int n = 0; n < 6; ++n) hole[n] = 0;
  return true;
}

// `staged`: asked by fdtdhip_step_stage, the whole step in pieces around the halo exchanges
// of an x-slab -- the marching step or nothing.
FusedKind fused_kind(const fdtdhip_plan* p, bool staged, double shell_share) {
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
  for (const DetH& d : p->dets)
    if (staged && d.below) return FUSED_NONE;   // read between the post pieces and a plane exchange
  // periodic boundaries: on undivided grids the marching steps take their planes into the shell
  // and copy by launches of their own; a PML registered after one is corrected after the copy,
  // which only the half-step path does
  const bool periodic = periodic_by_launch(p);
  for (const BoundaryRef& b : p->boundaries)
    if (b.kind == 1 && (b.idx > 2 || staged)) return FUSED_NONE;
  for (const PmlH& h : p->pmls)
    if (!h.psiE_alt || !h.fused) return FUSED_NONE;
  // sources: point lists that add, up to MAX_STEP_SOURCES points in the kernels' registers;
  // more of them, and boxes that are set, by launches of their own around the marching kernel
  // (not on slabs, whose ghost planes take launches of their own already)
  const bool apart = sources_by_launch(p);
  for (const SourceH& s : p->sources)
    if (!(s.kind == FDTDHIP_SRC_ADD_POINTS && !s.serial) && !(apart && s.kind == FDTDHIP_SRC_SET_BOX))
      return FUSED_NONE;
  if (apart && staged) return FUSED_NONE;
  const bool cell_ok = !apart && !periodic && p->nx <= 65535 && (int64_t)p->ny * p->nz < (1ll << 30) &&
                       p->cells() * 3 < (1ll << 31);   // the kernel's element offsets are 32-bit
  int core[6], hole[6] = {0, 0, 0, 0, 0, 0};
  const bool whole_grid = p->pmls.empty() && p->objs.empty() && !halo && !apart && !periodic;   // step_march, no shell
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
  if (!(p->mat_const[0] && p->mat_const[1]) && !(p->pmls.empty() && p->objs.empty() && !apart && !periodic))
    return FUSED_NONE;
  // (PML faces and objects: the shell and the box of the objects take half steps of their
  // own at a third of the marching kernel's rate per cell -- worth it while they are few)
  // (256^3 with six ten-cell faces, 22 % of the cells: float64 0.489 against 0.508 ms per step
  // in a bare loop and no difference in the benchmark, float32 0.256 against 0.235)
  // (size: 2^22 cells in float32 -- 160^3 with six ten-cell faces runs at 52.4 Gcells/s by half
  // steps and 44.5 two steps per pass, 192^3 at 61 either way --, 4.0 M in float64, where 160^3
  // gains: 29.0 -> 33.2, and 144^3 does not: 29.3 -> 28.9)
  // Whole grids (no shell at all) in float64 from 2.0 M: 128^3 40.0 -> 49.6, 112^3 32.7 -> 35.1,
  // 96^3 28.2 -> 23.4; in float32 144^3 loses (70.5 -> 64.2) and 160^3 gains (67.3 -> 70.8).
  // (With faces float32 is a close call up to 200^3 -- 176^3 55.8 by half steps against 51.3,
  // 192^3 66.9 / 60.9, 200^3 56.7 / 62.8, 224^3 58.9 / 75.8: rows that are whole multiples of 64
  // cells favour the half-step kernels.)
  const int64_t min_cells = p->dtype == FDTDHIP_F32 ? (1ll << 22) : (whole_grid ? 2000000 : 4000000);
  if (march_ok && p->nz >= 128 && p->ny >= 64 && p->nx >= 64 && p->cells() >= min_cells &&
      (double)shell_cells <= shell_share * (double)p->cells())
    return FUSED_MARCH;
  return FUSED_NONE;
}

namespace {

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

}  // namespace

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
// fullest rounds wins, counting the `extra` planes every chunk evaluates beyond its own (one
// for the one-step kernels), and a single round counting 3 % less than several (512^3 float32: 4 chunks of 128 planes in one round
// 1.215 ms, 8 chunks of 64 in two rounds 1.183).
int march_chunk_planes(int planes, int tiles, int slots, int extra) {
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
    const int64_t nchunks = (planes + lx - 1) / lx;
    const int64_t wgs = (int64_t)tiles * nchunks;
    // with 16 tiles or more they are dealt to the XCDs in bands of ceil(tiles / 8), each XCD
    // with an eighth of the places: the fullest XCD sets the number of rounds (82 tiles in
    // bands of 11 and 6 chunks: 66 workgroups on 32 places, three rounds where the whole
    // machine would count 1.92 -- the two-step kernel then took 1.78 instead of 1.43 ms)
    const int64_t rounds = tiles >= 16 ? (((tiles + 7) / 8) * nchunks + slots / 8 - 1) / std::max(slots / 8, 1)
                                       : (wgs + slots - 1) / slots;
    const double score = (double)wgs / (double)(rounds * slots) * lx / (lx + (double)extra) * (rounds >= 2 ? 1.0 : 0.97);
    if (score > best + 1e-9) { best = score; best_lx = lx; }
  }
  return std::max(best_lx, 1);
}

namespace {

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

template <typename T, int TY, int MW>
int launch_march_shell_tiles(fdtdhip_plan* p, const StepArgs<T>& a, bool obj, int stages, int64_t q) {
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
    if (periodic_by_launch(p))      // (the copy of the boundary planes, then the sources: the order of a half step)
      if (int rc = apply_periodic_to(p, true, a.E1)) return rc;
    if (ss_free.n) hipLaunchKernelGGL((shell_sources<T>), dim3(1), dim3(64), 0, p->stream, a.E1, ss_free);
    // (sources applied apart: all their cells lie in the shell or in the box inside the core that
    // the launch above stepped; the marching kernel reads their new E from the second buffer)
    if (sources_by_launch(p))
      if (int rc = apply_sources_to(p, true, q, a.E1)) return rc;
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
  if ((stages & ST_H_FREE) && periodic_by_launch(p))
    if (int rc = apply_periodic_to(p, false, a.H1)) return rc;
  if ((stages & ST_H_FREE) && sources_by_launch(p))      // (boxes that are set: H after its half step)
    if (int rc = apply_sources_to(p, false, q, a.H1)) return rc;
  HIPCHK(hipGetLastError());
  return 0;
}

template <typename T>
int launch_march_shell(fdtdhip_plan* p, const StepArgs<T>& a, bool obj, int stages, int64_t q) {
  if ((p->nz + 63) / 64 <= 8)
    return launch_march_shell_tiles<T, sizeof(T) == 4 ? FDTDHIP_MARCH_TY32 : FDTDHIP_MARCH_TY64, 8>(p, a, obj, stages, q);
  return launch_march_shell_tiles<T, sizeof(T) == 4 ? FDTDHIP_MARCH_TY32 : 2, MARCH_MAX_WAVES>(p, a, obj, stages, q);
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
    fill_object_override<T>(p, b);
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
  const bool apart = sources_by_launch(p);
  for (const SourceH& s : p->sources) {
    if (apart) break;      // (applied by launches of their own)
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
    return half_post(p, false, q, POST_FINISH | POST_SKIP_CELL_DETECTORS);
  }
  if (int rc = half_post(p, true, q, POST_FINISH)) return rc;
  return half_post(p, false, q, POST_FINISH);
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
  if (march && (pml || obj || sources_by_launch(p) || periodic_by_launch(p))) {
    if (int rc = launch_march_shell<T>(p, a, obj, ST_ALL, q)) return rc;
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
  if (int rc = launch_march_shell<T>(p, a, a.e.nobj > 0, pieces[stage], q)) return rc;
  return stage == 3 ? step_finish<T>(p, q, false) : 0;
}

}  // namespace

// ---- two steps per pass (yee_march2.hpp) ------------------------------------------------
// Whole grids without PML faces: one launch of the two-step kernel.  Plans with PML faces:
// the two-step kernel advances the box `inner` (pair_boxes: the PML-free core shrunk by one
// cell on its low sides and by two on its high sides, so that everything the cells' two steps
// depend on is plain and it needs the state before the pair and nothing else), and the cells
// outside `inner` take the two steps one after the other with the shell kernels of the one-step
// march (yee_shell.hpp):
//   1. two-step kernel over `inner`:            (E, H)(n) of A  ->  (E, H)(n+2) of B
//   2. first step of the others, in place in A (nothing reads A's old values any more), over a
//      few layers of `inner` as well (pair_boxes: e1, h1): their state after the first step,
//      which the two-step kernel kept in registers, is what the second step of the cells next
//      to `inner` reads; evaluated as the plain cells they are
//   3. second step of the cells outside `inner`, from A into B: E half step, then H half
//      step, which reads the new E of `inner`'s outermost cells from B.
// Same operations on the same operands as four half steps everywhere: bitwise equal.

constexpr double MARCH2_OUTSIDE_SHARE = 0.4;   // cells outside the inner box: at most this share

namespace {

// The core box of a plan shrunk by `lo` cells on every low side and `hi` cells on every high
// side where a shell lies (a side that coincides with a face of the grid has none).  False if
// nothing is left.
bool shrunk_core(const fdtdhip_plan* p, const int (&core)[6], int lo, int hi, int (&box)[6]) {
  const int n[3] = {p->nx, p->ny, p->nz};
  for (int ax = 0; ax < 3; ++ax) {
    box[2 * ax] = core[2 * ax] > 0 ? core[2 * ax] + lo : 0;
    box[2 * ax + 1] = core[2 * ax + 1] < n[ax] ? core[2 * ax + 1] - hi : n[ax];
    if (box[2 * ax + 1] - box[2 * ax] < 1) return false;
  }
  return true;
}

bool inside_box(const fdtdhip_plan* p, const int (&box)[6], int64_t elem) {
  const int64_t plane = (int64_t)p->ny * p->nz, cell = elem / 3;
  const int i = (int)(cell / plane), j = (int)((cell % plane) / p->nz), k = (int)(cell % p->nz);
  return i >= box[0] && i < box[1] && j >= box[2] && j < box[3] && k >= box[4] && k < box[5];
}

// The cells outside `c` as disjoint boxes: x slabs over everything, y slabs between them, z
// slabs inside.
// x_from / x_to: the planes below / from there on are left out (an open slab face's plane, which
// waits for its ghost plane and takes a launch of its own).
// island: one more box inside `c` (objects and the layers around them), or null.
Shell shell_around(const fdtdhip_plan* p, const int (&c)[6], int x_from, int x_to, const int* island = nullptr) {
  const int ny = p->ny, nz = p->nz;
  const ShellBox none{0, 0, 0, 0, 0, 0};
  const ShellBox boxes[7] = {{x_from, c[0], 0, ny, 0, nz}, {c[1], x_to, 0, ny, 0, nz},
                             {c[0], c[1], 0, c[2], 0, nz}, {c[0], c[1], c[3], ny, 0, nz},
                             {c[0], c[1], c[2], c[3], 0, c[4]}, {c[0], c[1], c[2], c[3], c[5], nz},
                             island ? ShellBox{island[0], island[1], island[2], island[3], island[4], island[5]} : none};
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
}

// `hole` grown by m cells on every side and cut to `box`; all zeros if nothing is left (or there is no hole).
void grown_hole(const int (&hole)[6], int lo, int hi, const int (&box)[6], int (&out)[6]) {
  bool any = hole[1] > hole[0];
  for (int ax = 0; ax < 3 && any; ++ax) {
    out[2 * ax] = std::max(hole[2 * ax] - lo, box[2 * ax]);
    out[2 * ax + 1] = std::min(hole[2 * ax + 1] + hi, box[2 * ax + 1]);
    any = out[2 * ax + 1] > out[2 * ax];
  }
  if (!any)
    for (int n = 0; n < 6; ++n) out[n] = 0;
}

// The boxes of a pair of steps.  The E half step reads H of a cell and of its neighbours
// BELOW it (x - e), the H half step E of the cell and of its neighbours ABOVE (reference
// fdtd/grid.py:42-49, 67-74), so what a cell's state after two steps depends on reaches one
// cell down and two cells up along every axis:
//   inner  the cells the two-step kernel stores: their boxes [-1, +2]^3 are plain cells, i.e. the
//          core shrunk by 1 on its low sides and by 2 on its high sides; O = the cells outside it;
//   h1     the first step's H half step runs outside it: O and the layer of `inner` under each of
//          its high faces, whose H' the second E half step of the cells above them reads;
//   e1     the first step's E half step runs outside it: the cells of the H half step and the
//          ones above each of them -- one layer of `inner` more on the low sides.
// With objects (or sources applied by launch) the box that holds them, grown by the mirror
// image of those margins and clipped, is an island inside each box that belongs to the cells
// outside it.  False: the plan has no room for them.
struct PairBoxes { int inner[6], e1[6], h1[6], hole2[6], hole4[6], hole3[6]; };
bool pair_boxes(const fdtdhip_plan* p, PairBoxes& b) {
  int core[6] = {0, p->nx, 0, p->ny, 0, p->nz}, hole[6] = {0, 0, 0, 0, 0, 0};
  if ((!p->pmls.empty() || !p->objs.empty() || p->H_below || p->E_above || sources_by_launch(p) ||
       periodic_by_launch(p)) &&
      !march_core(p, core, hole))
    return false;
  if (!(shrunk_core(p, core, 1, 2, b.inner) && shrunk_core(p, core, 1, 3, b.h1) &&
        shrunk_core(p, core, 2, 3, b.e1)))
    return false;
  grown_hole(hole, 2, 1, b.inner, b.hole2);
  grown_hole(hole, 3, 1, b.h1, b.hole3);
  grown_hole(hole, 3, 2, b.e1, b.hole4);
  return true;
}

// The pieces of a pair of steps.  On an x-slab two planes wait for a neighbour's plane in every
// step -- plane 0 in the E half step, the last plane in the H half step -- and take launches of
// their own (..._GHOST), so that a caller can place the rest around the four plane exchanges
// of the pair.  The two-step kernel reads no ghost plane at all (its box keeps two planes from an
// open low face and three from an open high one) and comes first: the first step of the others overwrites what it reads.
enum { PST_INNER = 1, PST_E1_FREE = 2, PST_E1_GHOST = 4, PST_H1_FREE = 8, PST_H1_GHOST = 16,
       PST_E2_FREE = 32, PST_E2_GHOST = 64, PST_H2_FREE = 128, PST_H2_GHOST = 256, PST_ALL = 511 };

template <typename T, int TY, int TYR>
int pair_step(fdtdhip_plan* p, int64_t q, int pieces) {
  March2Args<T> m;
  m.E0 = (const T*)p->E; m.H0 = (const T*)p->H;
  m.E1 = (T*)p->E_alt;   m.H1 = (T*)p->H_alt;
  m.nx = p->nx; m.ny = p->ny; m.nz = p->nz;
  PairBoxes pb;
  if (!pair_boxes(p, pb)) return fail(-1, "internal: no room for two steps per pass");
  for (int n = 0; n < 6; ++n) { m.box[n] = pb.inner[n]; m.hole[n] = pb.hole2[n]; }
  const int (&e1)[6] = pb.e1;
  const int (&h1)[6] = pb.h1;
  const bool obj = !p->objs.empty();
  const bool periodic = periodic_by_launch(p);
  const bool shell = !p->pmls.empty() || obj || p->H_below || p->E_above || sources_by_launch(p) || periodic;
  const int lo_skip = p->H_below ? 1 : 0, hi_skip = p->E_above ? 1 : 0;
  const int64_t plane3 = (int64_t)p->ny * p->nz * 3;
  // the products the one-step kernels form per cell, (Courant number x inverse material),
  // rounded once in T here as there
  StepArgs<T> a;
  if (int rc = step_args<T>(p, q, true, a)) return rc;
  if (!a.e.mat_const || !a.h.mat_const) return fail(-1, "internal: two steps per pass need constant materials");
  const T sc = (T)p->sc;
  for (int c = 0; c < 3; ++c) { m.ce[c] = sc * a.e.bg[c]; m.ch[c] = sc * a.h.bg[c]; }
  m.nsrc = 0;
  // the source points outside `inner`, per step; those on plane 0 of a slab open below apart
  ShellSources<T> s1, s2, g1, g2;
  s1.n = s2.n = g1.n = g2.n = 0;
  // (sources applied apart: by launches of their own after each half step of the cells that the
  // shell launches step -- all their cells are among those)
  const bool apart = sources_by_launch(p);
  for (const SourceH& s : p->sources) {
    if (apart) break;
    const double w0 = waveform(s, q), w1 = waveform(s, q + 1);
    for (int64_t i = 0; i < s.n; ++i) {
      if (m.nsrc >= MAX_STEP_SOURCES) return fail(-1, "internal: too many source points for a whole step");
      const int64_t cell = s.elem_host[i] / 3, plane = (int64_t)p->ny * p->nz;
      m.src_i[m.nsrc] = (int)(cell / plane);
      m.src_j[m.nsrc] = (int)((cell % plane) / p->nz);
      m.src_k[m.nsrc] = (int)(cell % p->nz);
      m.src_c[m.nsrc] = (int)(s.elem_host[i] % 3);
      m.src_v[0][m.nsrc] = (T)(s.profile_host[i] * w0);
      m.src_v[1][m.nsrc] = (T)(s.profile_host[i] * w1);
      const bool ghost = lo_skip && s.elem_host[i] < plane3;
      if (shell && (!inside_box(p, e1, s.elem_host[i]) || inside_box(p, pb.hole4, s.elem_host[i]))) {
        ShellSources<T>& ss = ghost ? g1 : s1;
        ss.elem[ss.n] = s.elem_host[i];
        ss.value[ss.n++] = m.src_v[0][m.nsrc];
      }
      if (shell && (!inside_box(p, m.box, s.elem_host[i]) || inside_box(p, pb.hole2, s.elem_host[i]))) {
        ShellSources<T>& ss = ghost ? g2 : s2;
        ss.elem[ss.n] = s.elem_host[i];
        ss.value[ss.n++] = m.src_v[1][m.nsrc];
      }
      ++m.nsrc;
    }
  }
  for (int n = m.nsrc; n < MAX_STEP_SOURCES; ++n) {
    m.src_i[n] = m.src_j[n] = m.src_k[n] = m.src_c[n] = -1;
    m.src_v[0][n] = m.src_v[1][n] = T(0);
  }
  // detectors: the first step's readings of the cells the two-step kernel stores are evaluated
  // once more by a kernel of their own, from the state before the pair (yee_march2.hpp)
  March2Dets<T> dets;
  dets.count = 0;
  dets.first[0] = 0;
  for (const DetH& d : p->dets) {
    if (d.kind != FDTDHIP_DET_CELLS || d.n == 0) continue;
    if (dets.count >= MARCH2_MAX_DETS) return fail(-1, "internal: more detectors than two steps per pass serve");
    // (two rows per pair; a call for a later piece of a pair in stages finds the first one taken)
    if ((pieces & PST_INNER) && (!d.outE || !d.outH || d.usedE + 2 > d.capacity || d.usedH + 2 > d.capacity))
      return fail(-3, "detector output capacity exhausted");
    dets.cell[dets.count] = d.cell;
    dets.outE[dets.count] = (T*)d.outE + d.usedE * d.n * 3;
    dets.outH[dets.count] = (T*)d.outH + d.usedH * d.n * 3;
    dets.first[dets.count + 1] = dets.first[dets.count] + d.n;
    ++dets.count;
  }
  // z rows of up to 512 cells: a workgroup of up to eight waves per tile, which hand each other
  // what crosses their ends through LDS (step_march2_rows); longer rows: independent waves
  // (step_march2).  FDTDHIP_MARCH2_ROWS=0 in the environment: independent waves always (tests).
  const int waves = (p->nz + 63) / 64;
  const char* env = getenv("FDTDHIP_MARCH2_ROWS");
  // (float32 rows of up to 1024 cells: 16 waves with two-row tiles)
  const bool rows16 = sizeof(T) == 4 && waves > MARCH2_ROW_WAVES && waves <= 16 && !(env && atoi(env) == 0);
  const bool rows = (waves <= MARCH2_ROW_WAVES && !(env && atoi(env) == 0)) || rows16;
  const int planes = m.box[1] - m.box[0], nrows = m.box[3] - m.box[2], ncells = m.box[5] - m.box[4];
  void (*kernel)(March2Args<T>);
  int groups, threads, extra;
  if (rows16) {
    m.tiles = (nrows + 1) / 2;
    m.nseg = waves;
    groups = m.tiles;
    threads = 64 * waves;
    extra = 4;
    if constexpr (sizeof(T) == 4) kernel = step_march2_rows<T, 2, 16>;
    else return fail(-1, "internal: no 16-wave two-step kernel in float64");
  } else if (rows) {
    m.tiles = (nrows + TYR - 1) / TYR;
    m.nseg = waves;
    groups = m.tiles;
    threads = 64 * waves;
    extra = 4;
    kernel = step_march2_rows<T, TYR>;
  } else {
    m.tiles = (nrows + TY - 1) / TY;
    m.nseg = (ncells + MARCH2_ZOWN - 1) / MARCH2_ZOWN;
    groups = (m.tiles * m.nseg + MARCH2_WAVES - 1) / MARCH2_WAVES;
    threads = 64 * MARCH2_WAVES;
    extra = 3;
    kernel = step_march2<T, TY>;
  }
  m.lx = march_chunk_planes(planes, groups, march_slots(p, (const void*)kernel, threads), extra);
  m.band = groups >= 16 ? (groups + 7) / 8 : 0;
  const dim3 grid(m.band ? 8 * m.band : groups, (planes + m.lx - 1) / m.lx, 1), block(threads);
  size_t slot;
  if (pieces & PST_INNER) {
    if (int rc = span_begin(p, slot)) return rc;
    hipLaunchKernelGGL(kernel, grid, block, 0, p->stream, m);
    HIPCHK(hipGetLastError());
    if (int rc = span_end(p, slot, TIME_PAIR)) return rc;
    ++two_step_launches;
    if (dets.count > 0) {
      hipLaunchKernelGGL((march2_detect_first_step<T>), dim3(blocks_for(dets.first[dets.count])), dim3(256), 0,
                         p->stream, m, dets);
      HIPCHK(hipGetLastError());
    }
  }
  T* EA = (T*)p->E; T* HA = (T*)p->H; T* EB = (T*)p->E_alt; T* HB = (T*)p->H_alt;
  const bool pml = !p->pmls.empty();
    // ghost = false: the cells outside `in` that wait for no ghost plane; true: the plane that does
    auto half_step = [&](bool phaseE, bool ghost, const int (&in)[6], const int (&island)[6],
                         const T* F, const T* G, T* out) -> int {
      Shell sh;
      if (ghost) {      // plane 0 (E) or the last plane (H), whole
        const int x = phaseE ? 0 : p->nx - 1;
        const int none[6] = {x, x + 1, 0, 0, 0, 0};      // (no rows: the y slab above it is the plane)
        sh = shell_around(p, none, x, x + 1);
      } else {
        sh = shell_around(p, in, phaseE ? lo_skip : 0, phaseE ? p->nx : p->nx - hi_skip,
                          island[1] > island[0] ? island : nullptr);
      }
      if (!sh.count) return 0;
      BulkArgs<T> b = phaseE ? a.e : a.h;
      b.F = const_cast<T*>(F);
      b.G = G;
      // (recorded runs: the E half steps of the objects' box write the curl of H into the
      // current row of the objects' curl traces, which moves on after each step of the pair)
      b.trace_step = phaseE ? p->trace_cursor : -1;
      if (int rc = span_begin(p, slot)) return rc;
      const dim3 sgrid((unsigned)sh.first[sh.count]), sblock(256);
      if (phaseE && pml && obj) hipLaunchKernelGGL((shell_E<T, true, true>), sgrid, sblock, 0, p->stream, b, out, sh);
      else if (phaseE && pml) hipLaunchKernelGGL((shell_E<T, true, false>), sgrid, sblock, 0, p->stream, b, out, sh);
      else if (phaseE && obj) hipLaunchKernelGGL((shell_E<T, false, true>), sgrid, sblock, 0, p->stream, b, out, sh);
      else if (phaseE) hipLaunchKernelGGL((shell_E<T, false, false>), sgrid, sblock, 0, p->stream, b, out, sh);
      else if (pml) hipLaunchKernelGGL((shell_H<T, true>), sgrid, sblock, 0, p->stream, b, out, sh);
      else hipLaunchKernelGGL((shell_H<T, false>), sgrid, sblock, 0, p->stream, b, out, sh);
      HIPCHK(hipGetLastError());
      return span_end(p, slot, TIME_SHELL);
    };
    auto add = [&](T* E, const ShellSources<T>& ss) {
      if (ss.n) hipLaunchKernelGGL((shell_sources<T>), dim3(1), dim3(64), 0, p->stream, E, ss);
    };
  if (shell) {
    // first step, in place in A
    if (pieces & PST_E1_FREE) {
      if (int rc = half_step(true, false, e1, pb.hole4, EA, HA, EA)) return rc;
      if (periodic) if (int rc = apply_periodic_to(p, true, EA)) return rc;
      add(EA, s1);
      if (apart) if (int rc = apply_sources_to(p, true, q, EA)) return rc;
    }
    if ((pieces & PST_E1_GHOST) && lo_skip) { if (int rc = half_step(true, true, e1, pb.hole4, EA, HA, EA)) return rc; add(EA, g1); }
    if (pieces & PST_H1_FREE) {
      if (int rc = half_step(false, false, h1, pb.hole3, HA, EA, HA)) return rc;
      if (periodic) if (int rc = apply_periodic_to(p, false, HA)) return rc;
      if (apart) if (int rc = apply_sources_to(p, false, q, HA)) return rc;
    }
    if ((pieces & PST_H1_GHOST) && hi_skip) if (int rc = half_step(false, true, h1, pb.hole3, HA, EA, HA)) return rc;
    if ((pieces & PST_H1_GHOST) && dets.count > 0) {
      // the first step's readings of the other cells: from A, which holds their state after it
      DetRegion reg;
      reg.ny = p->ny; reg.nz = p->nz;
      for (int n = 0; n < 6; ++n) { reg.inner[n] = m.box[n]; reg.hole[n] = m.hole[n]; }
      for (int phase = 0; phase < 2; ++phase) {
        DetBatch<T> batch;
        batch.count = 0;
        batch.first[0] = 0;
        for (const DetH& d : p->dets) {
          if (d.kind != FDTDHIP_DET_CELLS || d.n == 0) continue;
          batch.cell[batch.count] = d.cell;
          batch.out[batch.count] = phase == 0 ? (T*)d.outE + d.usedE * d.n * 3 : (T*)d.outH + d.usedH * d.n * 3;
          batch.first[batch.count + 1] = batch.first[batch.count] + d.n * 3;
          ++batch.count;
        }
        hipLaunchKernelGGL((detect_cells_batch_outside<T>), dim3(blocks_for(batch.first[batch.count])), dim3(256), 0,
                           p->stream, phase == 0 ? (const T*)EA : (const T*)HA, batch, reg);
      }
      HIPCHK(hipGetLastError());
    }
  }
  if (pieces & PST_H1_GHOST) {
    for (DetH& d : p->dets)
      if (d.kind == FDTDHIP_DET_CELLS) { ++d.usedE; ++d.usedH; }      // the first step's record row
    advance_trace_cursor(p);                                          // ... and its row of the curl traces
  }
  if (shell) {
    // second step, from A into B
    if (pieces & PST_E2_FREE) {
      if (int rc = half_step(true, false, m.box, pb.hole2, EA, HA, EB)) return rc;
      if (periodic) if (int rc = apply_periodic_to(p, true, EB)) return rc;
      add(EB, s2);
      if (apart) if (int rc = apply_sources_to(p, true, q + 1, EB)) return rc;
    }
    if ((pieces & PST_E2_GHOST) && lo_skip) { if (int rc = half_step(true, true, m.box, pb.hole2, EA, HA, EB)) return rc; add(EB, g2); }
    if (pieces & PST_H2_FREE) {
      if (int rc = half_step(false, false, m.box, pb.hole2, HA, EB, HB)) return rc;
      if (periodic) if (int rc = apply_periodic_to(p, false, HB)) return rc;
      if (apart) if (int rc = apply_sources_to(p, false, q + 1, HB)) return rc;
    }
    if ((pieces & PST_H2_GHOST) && hi_skip) if (int rc = half_step(false, true, m.box, pb.hole2, HA, EB, HB)) return rc;
    HIPCHK(hipGetLastError());
  }
  if (pieces & PST_H2_GHOST) {
    std::swap(p->E, p->E_alt);
    std::swap(p->H, p->H_alt);
    p->step_flipped = !p->step_flipped;
    // the second step's readings: every cell's state is in memory (the E half's finish moves
    // the row of the curl traces on)
    if (int rc = half_post(p, true, q + 1, POST_FINISH)) return rc;
    if (int rc = half_post(p, false, q + 1, POST_FINISH)) return rc;
  }
  return 0;
}

}  // namespace

// Whether runs of this plan may advance two steps per pass: a marching plan, both materials
// constant, detectors of the cell kind only and few of them, room for two more rows in the
// curl traces of a recorded run, and room for the boxes of pair_step between the PML faces,
// the objects' box and the open slab faces (with material arrays to read twice the kernel is
// slower than two one-step launches: 512^3 float32 2.26 against 1.81 ms per step).  FDTDHIP_MARCH2=0 in the
// environment: never (read at every call: tests compare the two ways of stepping in one process).
bool whole_pair_possible(const fdtdhip_plan* p, int32_t inner_out[6], bool staged) {
  const char* e = getenv("FDTDHIP_MARCH2");
  if (e && atoi(e) == 0) return false;
  PairBoxes pb;
  const int (&inner)[6] = pb.inner;
  const bool halo = p->H_below || p->E_above;
  int served = 0;
  for (const DetH& d : p->dets) {      // detectors: cell lists, a few
    if (d.kind != FDTDHIP_DET_CELLS) return false;
    if (d.n > 0 && ++served > MARCH2_MAX_DETS) return false;
  }
  // (a recorded run: room for the two rows of a pair in every object's curl trace)
  if (trace_rows_left(p) < 2) return false;
  if (!(halo == staged &&
        p->nz >= MARCH2_ZOWN && p->mat_const[0] && p->mat_const[1] &&
        (int64_t)p->ny * p->nz * 3 * (p->dtype == FDTDHIP_F32 ? 4 : 8) < (1ll << 32) &&   // in-plane byte offsets
        pair_boxes(p, pb)))
    return false;
  // the marching kernels' other conditions and the size from which they pay, with the share of
  // the cells outside the inner box in place of the one-step march's shell: two steps per pass
  // pay up to a larger share (256^3 float64 with six ten-cell faces, 26 % outside the inner
  // box: 0.402 ms per step against 0.505 for half steps and 0.489 for the one-step march)
  const int64_t in = (int64_t)(inner[1] - inner[0]) * (inner[3] - inner[2]) * (inner[5] - inner[4]) -
                     (int64_t)(pb.hole2[1] - pb.hole2[0]) * (pb.hole2[3] - pb.hole2[2]) * (pb.hole2[5] - pb.hole2[4]);
  if (fused_kind(p, staged, 1.0) != FUSED_MARCH) return false;
  if (fused_kind(p, staged, 0.0) != FUSED_MARCH) {      // (not forced by FDTDHIP_FUSED=2, not a whole grid)
    if ((double)(p->cells() - in) > MARCH2_OUTSIDE_SHARE * (double)p->cells()) return false;
    // undivided float32 grids with faces: up to 192^3 the half-step kernels are as fast or
    // faster (176^3 55.8 against 51.3 Gcells/s, 192^3 66.9 / 60.9; 200^3 56.7 / 62.8)
    if (!staged && p->dtype == FDTDHIP_F32 && p->cells() < 7500000) return false;
  }
  if (inner_out)
    for (int n = 0; n < 6; ++n) inner_out[n] = inner[n];
  return true;
}

int whole_pair_step(fdtdhip_plan* p, int64_t q) {
  return DISPATCH(p, (pair_step<float, FDTDHIP_MARCH2_TY32, FDTDHIP_MARCH2_ROWS_TY32>(p, q, PST_ALL)),
                  (pair_step<double, FDTDHIP_MARCH2_TY64, FDTDHIP_MARCH2_ROWS_TY64>(p, q, PST_ALL)));
}

// One of the eight pieces of a pair of steps on an x-slab (fdtdhip_pair_stage).
int whole_pair_stage(fdtdhip_plan* p, int64_t q, int stage) {
  static const int pieces[8] = {PST_INNER | PST_E1_FREE, PST_E1_GHOST, PST_H1_FREE, PST_H1_GHOST,
                                PST_E2_FREE, PST_E2_GHOST, PST_H2_FREE, PST_H2_GHOST};
  return DISPATCH(p, (pair_step<float, FDTDHIP_MARCH2_TY32, FDTDHIP_MARCH2_ROWS_TY32>(p, q, pieces[stage])),
                  (pair_step<double, FDTDHIP_MARCH2_TY64, FDTDHIP_MARCH2_ROWS_TY64>(p, q, pieces[stage])));
}

int whole_fused_step(fdtdhip_plan* p, int64_t q, FusedKind kind) {
  return DISPATCH(p, fused_step<float>(p, q, kind), fused_step<double>(p, q, kind));
}

int whole_staged_step(fdtdhip_plan* p, int64_t q, int stage) {
  return DISPATCH(p, staged_step<float>(p, q, stage), staged_step<double>(p, q, stage));
}

}  // namespace fdtdhip
