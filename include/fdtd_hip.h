/*
 * fdtd_hip.h -- C ABI of the MI355X (gfx950) Yee time-stepping engine.
 *
 * The reference (simenhu/fdtd) is pure Python and has no FFI; its seam for a
 * new device is the array backend swapped by set_backend() (reference
 * fdtd/backend.py:354-426) together with the half-step methods of the Grid
 * (reference fdtd/grid.py:250-325).  Every entry point below names the piece of
 * reference code it stands in for.
 *
 * Conventions
 *   - All functions return 0 on success and a negative code on failure;
 *     fdtdhip_last_error() returns the message of the last failure on the
 *     calling thread.
 *   - Field memory is CALLER OWNED device memory, C-contiguous
 *     (nx, ny, nz, 3) arrays of the plan's element type, the layout of the
 *     reference's Grid.E / Grid.H / Grid.inverse_permittivity /
 *     Grid.inverse_permeability (reference fdtd/grid.py:135-151).  The plan
 *     only borrows the pointers.
 *   - Small read-only tables (PML profiles, index lists, object descriptors)
 *     are copied into device memory owned by the plan when they are added.
 *   - A plan is not thread safe.  All work is enqueued on the plan's stream
 *     (fdtdhip_plan_set_stream).  The time-stepping calls do not wait for the
 *     device: descriptor tables that changed (objects or PMLs added, buffers
 *     re-pointed) are uploaded by stream-ordered copies out of pinned staging
 *     buffers.  fdtdhip_plan_destroy, fdtdhip_plan_reset, fdtdhip_plan_scan_materials,
 *     fdtdhip_plan_get_timing and fdtdhip_adjoint_run_noalloc wait for the stream.
 */
#ifndef FDTD_HIP_H
#define FDTD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDTDHIP_ABI_VERSION 1

/* element types (reference: backend.float, fdtd/backend.py:403,415) */
#define FDTDHIP_F32 0
#define FDTDHIP_F64 1

/* object kinds (reference fdtd/objects.py: Object 27, AbsorbingObject 160,
 * AnisotropicObject 229 / LearnableAnisotropicObject 276 with is_substrate=False) */
#define FDTDHIP_OBJ_ISOTROPIC 0
#define FDTDHIP_OBJ_ABSORBING 1
#define FDTDHIP_OBJ_ANISOTROPIC 2
#define FDTDHIP_OBJ_ANISOTROPIC_MODULATED 3

/* source kinds (reference fdtd/sources.py: PointSource 26 / LineSource 170 add to
 * one component at a list of cells; PlaneSource 359 overwrites a box) */
#define FDTDHIP_SRC_ADD_POINTS 0
#define FDTDHIP_SRC_SET_BOX 1
#define FDTDHIP_SRC_MODULATED_BOX 2 /* registered with fdtdhip_plan_add_modulated_source */
#define FDTDHIP_SRC_SOFT_POINT 3    /* registered with fdtdhip_plan_add_soft_source */

/* detector kinds (reference fdtd/detectors.py: LineDetector 20 and BlockDetector
 * 146 copy the three components of listed cells; CurrentDetector 284) */
#define FDTDHIP_DET_CELLS 0
#define FDTDHIP_DET_CURRENT 1

typedef struct fdtdhip_plan fdtdhip_plan;

int fdtdhip_abi_version(void);
const char* fdtdhip_last_error(void);

/* Number of visible HIP devices (0 when there is none); stands in for the
 * availability check of set_backend (reference fdtd/backend.py:378-385). */
int fdtdhip_device_count(void);

/* One plan per Grid (reference fdtd/grid.py:90-169).  `courant` is
 * Grid.courant_number; (nx, ny, nz) is the extent of the arrays this plan works
 * on (one x-slab of a decomposed grid, or the whole grid). */
int fdtdhip_plan_create(fdtdhip_plan** out, int nx, int ny, int nz, int dtype,
                        double courant, int device);
int fdtdhip_plan_destroy(fdtdhip_plan* plan);

/* Forget every boundary, object, source and detector (used when the Python
 * lists of the Grid changed, reference fdtd/grid.py:157-166). */
int fdtdhip_plan_reset(fdtdhip_plan* plan);

/* HIP stream (hipStream_t) on which all later work of the plan is enqueued. */
int fdtdhip_plan_set_stream(fdtdhip_plan* plan, void* stream);

/* Borrow the four field arrays (reference fdtd/grid.py:135-151). */
int fdtdhip_plan_set_fields(fdtdhip_plan* plan, void* E, void* H,
                            const void* inv_permittivity, const void* inv_permeability);

/* x-slab decomposition (no reference counterpart: the reference is single
 * device).  Only two components cross an x face: update_E at plane i reads Hy, Hz
 * of plane i-1 (reference fdtd/grid.py:71,73), update_H at plane i reads Ey, Ez of
 * plane i+1 (fdtd/grid.py:46,48).  `H_below`: device (ny,nz,2) array holding (Hy, Hz)
 * of the neighbour's last plane, used by update_E for plane 0 instead of the
 * domain-edge mask.  `E_above`: (Ey, Ez) of the neighbour's first plane, used by
 * update_H for the last plane.  NULL = true domain edge. */
int fdtdhip_plan_set_halo(fdtdhip_plan* plan, const void* H_below, const void* E_above);

/* fdtdhip_plan_set_stream, fdtdhip_plan_set_fields, fdtdhip_plan_set_step_buffers and
 * fdtdhip_plan_set_halo in one call, for hosts that bind everything before every step and pay
 * for each crossing of the boundary (a loop that records one step per tape node, reference
 * cortical/autoencoder.py:108-112).  NULL for the optional arrays as in the single calls. */
int fdtdhip_plan_bind(fdtdhip_plan* plan, void* stream, void* E, void* H,
                      const void* inv_permittivity, const void* inv_permeability, void* E_alt,
                      void* H_alt, const void* H_below, const void* E_above);

/* PML slab (reference fdtd/boundaries.py:356-476).  axis 0..2, side 0 = low /
 * 1 = high, `thickness` cells.  bE,cE,bH,cH: HOST arrays of `thickness`
 * doubles, the normal-axis component of the reference's bE/cE/bH/cH
 * (boundaries.py:384-396).  psiE / psiH: caller-owned, zero-initialised device
 * arrays of shape (2, slab cells): the two recursive-convolution memory
 * variables that can be non-zero for this face, [0] = psi_{E,u}[..., axis],
 * [1] = psi_{E,v}[..., axis] with u = (axis+1)%3, v = (axis+2)%3
 * (boundaries.py:437-444, 465-472).  Boundaries act in the order added
 * (grid.py:279-291).  Returns the PML id (>= 0). */
int fdtdhip_plan_add_pml(fdtdhip_plan* plan, int axis, int side, int thickness,
                         const double* bE, const double* cE, const double* bH,
                         const double* cH, void* psiE, void* psiH);

/* Point PML `id` at other memory-variable arrays of the same shape (a caller that keeps
 * the state of a run in fresh arrays, as a recorded differentiable run does -- the
 * reference's autograd likewise makes every in-place update a new tensor version,
 * fdtd/boundaries.py:428-448).  Cheaper than rebuilding the plan: only the descriptor
 * tables are uploaded again before the next step. */
int fdtdhip_plan_set_pml_state(fdtdhip_plan* plan, int id, void* psiE, void* psiH);

/* Periodic boundary along `axis` (reference fdtd/boundaries.py:173-208); axis may be
 * FDTDHIP_PERIODIC_X_EXTERNAL (see fdtdhip_update_E_post_stage).  Along an axis of one
 * cell the copy maps a plane onto itself (boundaries.py:177, 182 with index 0 == -1); it
 * is accepted and not recorded. */
#define FDTDHIP_PERIODIC_X_EXTERNAL 3
int fdtdhip_plan_add_periodic(fdtdhip_plan* plan, int axis);

/* Object occupying box = {x0,x1,y0,y1,z0,z1} (reference fdtd/objects.py:116-126,
 * 204-218, 251-266).  inv: device (bx,by,bz,3), or (bx,by,bz,3,3) for
 * FDTDHIP_OBJ_ANISOTROPIC.  absorb: device (bx,by,bz,3) absorption factor
 * (objects.py:195-202) for FDTDHIP_OBJ_ABSORBING, else NULL.  The caller has
 * already zeroed the grid's inverse permittivity in the box (objects.py:90).
 * Returns the object id (>= 0). */
int fdtdhip_plan_add_object(fdtdhip_plan* plan, int kind, const int32_t box[6],
                            const void* inv, const void* absorb);

/* Source.  FDTDHIP_SRC_ADD_POINTS: cells = HOST array of 3*n ints (x,y,z per
 * point), profile = HOST array of n doubles; each step E[cell, comp_E] +=
 * profile[i]*w(q) (reference fdtd/sources.py:95-111, 316-337).
 * FDTDHIP_SRC_SET_BOX: cells = 6 ints {x0,x1,y0,y1,z0,z1}, n = 1,
 * profile[0] = amplitude; E[box, comp_E] = amplitude*w(q) in update_E and
 * H[box, comp_H] = amplitude*w(q) in update_H (sources.py:515-525).
 * w(q) = sin(2*pi*q/period + phase), or with pulse != 0 the Hanning pulse
 * hanning(frequency, q*hanning_dt, cycle) for q < int(2*pi/(frequency*hanning_dt/cycle)),
 * else 0 (sources.py:99-110; waveforms.py:8-9). */
int fdtdhip_plan_add_source(fdtdhip_plan* plan, int kind, int comp_E, int comp_H,
                            int64_t n, const int32_t* cells, const double* profile,
                            double period, double phase, int pulse, double frequency,
                            double hanning_dt, double cycle);

/* Energy-modulated additive source over box = {x0,x1,y0,y1,z0,z1} (reference
 * fdtd/sources.py:542-608, CorticalColumnPlaneSource): after the boundaries of every E
 * half step
 *     E[cell, comp] += sum_c osc[q - q0][c] * P[c][column]
 *                            * sigmoid(w[c][0]*sum(E^2) + w[c][1]*sum(H^2) + w[c][2])
 * with the energies of the cell's own fields at that moment and `column` the (x, y)
 * position in the box, x-major.  The reference's factors that do not depend on the
 * fields -- activation image, footprint of the normalised 3x3 direction kernels
 * (conv_transpose2d of ones, sources.py:599-602), amplitude -- are one device array
 * P (channels, columns); its oscillators sin(2*pi*t*f_c + phi_c) are a device table
 * osc (osc_rows, channels) for the steps q0 .. q0+osc_rows-1; weights is device
 * (channels, 3).  Until inputs are set (P = NULL) the source does nothing, as an
 * unseeded source in the reference.  Returns the source id.
 * Differentiable runs: energy_trace device (osc_rows, box cells, 2) is filled by the
 * forward pass; the reverse sweep then needs grad_P (channels, columns; accumulated),
 * grad_osc (osc_rows, channels; written per step) and grad_weights (channels, 3;
 * accumulated).  No gradient flows through the energies (sources.py:588-592). */
int fdtdhip_plan_add_modulated_source(fdtdhip_plan* plan, const int32_t box[6], int comp,
                                      int channels);
int fdtdhip_plan_set_modulated_source(fdtdhip_plan* plan, int id, const void* P, const void* osc,
                                      int64_t osc_rows, int64_t q0, const void* weights,
                                      void* energy_trace, void* grad_P, void* grad_osc,
                                      void* grad_weights);

/* Voltage source with a series resistance at one cell (reference fdtd/sources.py:612-737,
 * SoftArbitraryPointSource): after the boundaries of the E half step of step q
 *     v_in  = waveform[q]  (0 beyond its nwave samples)
 *     v_out = v_in + impedance * I   if impedance > 0, else v_in
 *     E[cell, z] += v_out / spacing
 * where I is the newest reading of the source's own loop-current detector (0 at q = 0):
 * the feedback never leaves the device.  waveform: HOST doubles (copied).  Returns the
 * source id.  fdtdhip_plan_link_soft_source names the FDTDHIP_DET_CURRENT detector (the
 * reference registers one at the same cell, sources.py:705-706) once it has been added.
 * fdtdhip_plan_set_soft_source_io, before a run: prev_current = device address of that
 * detector's last reading taken before this run (first point; NULL: none yet), and two
 * device arrays of the plan's element type with room for `capacity` steps that receive
 * v_in and v_out of every step (the reference's input_voltage / source_voltage lists).
 * Differentiable runs (the reference has the feedback on torch's tape: the current read at
 * sources.py:720 is a function of H): fdtdhip_plan_set_soft_source_adjoint, before
 * fdtdhip_adjoint_run / _cells, names the time step q0 of the adjoint's step 0 (at time step 0
 * there is no current); the adjoint of E at the cell then reaches the adjoint of H around
 * it.  The waveform is a constant (no gradient for it).  On an x-slab whose plane 0 holds the
 * cell the loop reaches into plane x - 1 of another rank (the one below; for plane 0 of the
 * simulation, wrapping around as the detector does, the last one):
 * fdtdhip_plan_set_soft_source_adjoint_below names a device (ny, nz, 3) plane to which the
 * terms of that plane are ADDED instead; the caller clears it, sends it to that rank after
 * fdtdhip_adjoint_cells(phase_E = 1) and the receiver adds it to the last plane of its lamH
 * before its fdtdhip_adjoint_gather.  NULL: an undivided grid (the wrap-around stays inside). */
int fdtdhip_plan_add_soft_source(fdtdhip_plan* plan, const int32_t cell[3], const double* waveform,
                                 int64_t nwave, double impedance, double spacing);
int fdtdhip_plan_link_soft_source(fdtdhip_plan* plan, int id, int detector_id);
int fdtdhip_plan_set_soft_source_io(fdtdhip_plan* plan, int id, const void* prev_current,
                                    void* vin_out, void* vout_out, int64_t capacity);
int fdtdhip_plan_set_soft_source_adjoint(fdtdhip_plan* plan, int id, int64_t q0);
int fdtdhip_plan_set_soft_source_adjoint_below(fdtdhip_plan* plan, int id, void* lamH_below);

/* Detector over n points.  FDTDHIP_DET_CELLS: cells = HOST 3*n ints; each
 * reading is (n,3) (reference fdtd/detectors.py:114-124, 241-265).
 * FDTDHIP_DET_CURRENT: each reading is (n) loop currents (detectors.py:443-462),
 * `spacing` = Grid.grid_spacing.  Returns the detector id (>= 0). */
int fdtdhip_plan_add_detector(fdtdhip_plan* plan, int kind, int64_t n,
                              const int32_t* cells, double spacing);

/* Where the next `capacity` readings of detector `id` go: device arrays of
 * capacity*(n*3) (outE, outH; FDTDHIP_DET_CELLS) or capacity*n (outH only;
 * FDTDHIP_DET_CURRENT) elements.  Stands in for the Python lists detector.E /
 * detector.H / detector.I (detectors.py:31-32, 319).  Each update_E / update_H
 * consumes one reading; running out of capacity is an error. */
int fdtdhip_plan_set_detector_output(fdtdhip_plan* plan, int id, void* outE, void* outH,
                                     int64_t capacity);

/* Loop-current detector on an x-slab (no reference counterpart: the reference is single
 * device).  The current of a cell reads H of plane x-1 (reference fdtd/detectors.py:452-459,
 * index -1 wrapping to the last plane of the simulation); for plane 0 of a slab that plane
 * is another rank's last one.  `H_plane`: device (ny,nz,3) copy of it.  A detector with such a
 * plane is left out of the update_H post pieces: the caller fills the plane once those have
 * run on the owning rank and then calls fdtdhip_read_deferred_currents, which takes one
 * reading of every such detector.  NULL restores the single-grid behaviour. */
int fdtdhip_plan_set_detector_below(fdtdhip_plan* plan, int id, const void* H_plane);
int fdtdhip_read_deferred_currents(fdtdhip_plan* plan);

/* Half steps (reference fdtd/grid.py:275-299 and 301-325); q =
 * Grid.time_steps_passed. */
int fdtdhip_update_E(fdtdhip_plan* plan, int64_t q);
int fdtdhip_update_H(fdtdhip_plan* plan, int64_t q);

/* The same half steps split for overlap with a halo exchange: `_bulk` is the
 * fused curl + material + object + PML pass over x planes [x0, x1); `_post`
 * is everything that follows it (periodic copies, PMLs registered after a
 * periodic boundary, sources, detectors). update_X = bulk(0,nx) + post. */
int fdtdhip_update_E_bulk(fdtdhip_plan* plan, int x0, int x1);
int fdtdhip_update_E_post(fdtdhip_plan* plan, int64_t q);
int fdtdhip_update_H_bulk(fdtdhip_plan* plan, int x0, int x1);
int fdtdhip_update_H_post(fdtdhip_plan* plan, int64_t q);

/* x-periodic boundary of an x-slab decomposition (reference fdtd/boundaries.py:173-184:
 * E[0] = E[-1] after the E update, H[-1] = H[0] after the H update): the two planes
 * live on the first and the last rank, so the copy is the caller's (a plane sent
 * between the two ranks).  Registered with fdtdhip_plan_add_periodic(plan,
 * FDTDHIP_PERIODIC_X_EXTERNAL) at its place in the boundary list -- later PMLs apply
 * their correction after it, as after any periodic boundary -- it makes the post
 * pieces come in two stages: stage 0 = the boundaries before it, [caller's copy],
 * stage 1 = the boundaries after it, sources, detectors. */
int fdtdhip_update_E_post_stage(fdtdhip_plan* plan, int64_t q, int stage);
int fdtdhip_update_H_post_stage(fdtdhip_plan* plan, int64_t q, int stage);

/* Second buffers for the fused full step (no reference counterpart: the reference
 * always runs the two half steps as separate passes, fdtd/grid.py:267-273).  When
 * E_alt / H_alt (shaped like E / H) and, for every PML, psiE_alt (shaped like psiE)
 * are provided, fdtdhip_run advances pairs of steps with one kernel per step (small grids:
 * no periodic boundary, no plane source, at most eight source points; large 3-D grids: the
 * cells next to PML faces, objects, periodic boundaries and large sources by launches beside
 * it), first -> second buffers and back; after the call the state is in the first buffers
 * again.  Contents of the second buffers are scratch.
 * Large 3-D plans with both material arrays proven constant by fdtdhip_plan_scan_materials,
 * sources that add at points without repeated cells or set a box (more than eight points
 * and boxes: undivided grids only), at most eight detectors, all
 * of the cell kind, advance TWO steps per pass over memory, in groups of four steps: the
 * state after the odd steps of a group then never exists in memory, except for the cells of
 * PML faces and of the box that holds the objects and the few layers next to them, which
 * take the two steps one after the other.  Detector readings of every step are delivered
 * as by half steps.  The result is the one of the half steps either way, bit for bit. */
int fdtdhip_plan_set_step_buffers(fdtdhip_plan* plan, void* E_alt, void* H_alt);
/* Whether fdtdhip_run would advance this plan two steps per pass (1) or not (0), and the box
 * {x0,x1,y0,y1,z0,z1} the two-step kernel covers (inner, may be NULL): the whole grid, or with
 * PML faces the PML-free box shrunk by one cell on every low side and by two on every high side
 * with a face (the reach of the Yee stencil over two steps); the cells outside it
 * (and those of the objects' box grown by two layers, which lies inside it) take the two steps
 * one after the other (no reference counterpart; benchmarks use it to
 * price the kernels). */
int fdtdhip_plan_two_step_box(fdtdhip_plan* plan, int staged, int32_t inner[6]);
/* How many launches of the two-step kernels this process has made so far, over all plans (no
 * reference counterpart; random-scene tests use it to tell which scenes took that path). */
int64_t fdtdhip_two_step_launches(void);
/* A pair of steps (time indices q, q + 1) of an x-slab in eight pieces around its four plane
 * exchanges (`staged` != 0 above asks whether the plan can take them), the counterpart of
 * fdtdhip_step_stage for two steps per pass.  With the fields in the first buffers A and the
 * second buffers B:
 *   stage 0  the two-step kernel over the inner box (A -> B; it keeps two planes from an open low
 *            face, three from an open high one, and reads no ghost plane), then the first step's E half step of the cells
 *            outside it except plane 0, in place in A          -- while H_below (step q) travels;
 *   stage 1  E half step of plane 0 (reads H_below).  Its new (Ey, Ez), in A, go to the rank below;
 *   stage 2  the first step's H half step except the last plane, in place in A;
 *   stage 3  H half step of the last plane (reads E_above).  Its new (Hy, Hz), in A, go to the rank above;
 *   stage 4  the second step's E half step except plane 0, A -> B;
 *   stage 5  E half step of plane 0 (reads H_below of step q + 1).  Its new (Ey, Ez), in B, go below;
 *   stage 6  the second step's H half step except the last plane;
 *   stage 7  H half step of the last plane (reads E_above of step q + 1); then the buffers swap.
 * Results equal four half steps bit for bit.  Take pairs in twos so that the state ends in the
 * first buffers.  The stages are accepted in their order only; a failed stage poisons the plan. */
int fdtdhip_pair_stage(fdtdhip_plan* plan, int64_t q, int stage);
int fdtdhip_plan_set_pml_step_buffer(fdtdhip_plan* plan, int id, void* psiE_alt);

/* nsteps full steps starting at time index q0 (reference fdtd/grid.py:250-273). */
int fdtdhip_run(fdtdhip_plan* plan, int64_t q0, int64_t nsteps);

/* Whole steps on an x-slab (no reference counterpart: the reference is single device and
 * always runs two passes per step, fdtd/grid.py:267-273).  With second buffers provided
 * (fdtdhip_plan_set_step_buffers, fdtdhip_plan_set_pml_step_buffer) a slab whose plan
 * fdtdhip_plan_whole_step_kind(plan, 1, ...) reports as 2 can advance a step in four pieces
 * placed around the two plane exchanges of the step; between the pieces the caller moves
 * the planes:
 *   stage 0  E half step of the cells next to PMLs, objects and the high slab face, and the
 *            first half of the whole-step kernel over the other cells -- needs neither ghost plane;
 *   stage 1  E half step of plane 0, which reads H_below: the plane must have arrived.  The
 *            new (Ey, Ez) of plane 0 are then in the second E buffer for the rank below;
 *   stage 2  second half of the whole-step kernel, H half step of the shell without the last plane;
 *   stage 3  H half step of the last plane, which reads E_above (arrived); then the buffers
 *            swap (the second buffers hold the state, as after an odd number of fused
 *            steps inside fdtdhip_run) and the detectors are read.
 * Results equal fdtdhip_update_E + fdtdhip_update_H bit for bit.  Take steps in pairs so
 * that the state ends in the first buffers.  The stages are accepted in their order only, and
 * while a step is in progress the other stepping calls (fdtdhip_update_*, fdtdhip_run) are
 * refused; a failed stage poisons the plan (reset it).
 * fdtdhip_plan_whole_step_kind: which whole-step kernel the plan would take -- 0 none, 1 the
 * one-lane-per-cell kernel (small 2-D grids), 2 the marching kernel; `staged` != 0 asks for
 * fdtdhip_step_stage (open slab faces allowed).  core / hole (may be NULL) receive the box
 * {x0,x1,y0,y1,z0,z1} the marching kernel advances and the box inside it whose E half step
 * it leaves to the shell launches (the objects). */
int fdtdhip_plan_whole_step_kind(fdtdhip_plan* plan, int staged, int32_t core[6], int32_t hole[6]);
int fdtdhip_step_stage(fdtdhip_plan* plan, int64_t q, int stage);

/* One whole step OUT OF PLACE (no reference counterpart: the reference's torch backend gets
 * new arrays from every operation it records, fdtd/grid.py:283-325 under autograd): reads the
 * bound fields, writes the new E and H into the step buffers and leaves the bound fields as
 * they were -- for callers that need the old state intact, such as a differentiable run
 * advanced one step per autograd node, which would otherwise copy both fields first.
 * Afterwards the roles are exchanged: the former step buffers are the plan's fields, the
 * former fields its step buffers; with the one-lane-per-cell kernel (return value 1) likewise
 * psi_E of every PML and its step buffer, while the marching kernel (2) and psi_H always
 * advance the memory variables in place.  Fails, without stepping, if the plan cannot take a
 * whole step in one pass (fdtdhip_plan_whole_step_kind returns 0). */
int fdtdhip_step_into(fdtdhip_plan* plan, int64_t q);

/* ---- differentiable runs (backward through time) ---------------------------
 * The reference differentiates through grid.run with torch autograd, recording
 * every slice operation of every step (reference examples/learnable_focus.py:98-128,
 * fdtd/objects.py:315-349).  Here the time stepping is linear in the fields, so
 * the backward pass is the adjoint recursion run by hand-written reverse-time
 * kernels; the only forward data it needs is the curl of H inside learnable
 * anisotropic objects, one (box,3) row per step.
 *
 * fdtdhip_plan_set_object_trace: device array of capacity*(box*3) elements in which
 *   the following E half steps record that curl for object `id` (row 0 first);
 *   NULL stops recording.
 * fdtdhip_plan_set_object_grad: device array shaped like the object's `inv`
 *   (box,3,3) to which the adjoint run ADDS dLoss/d(inv).
 * fdtdhip_plan_set_pml_adjoint: device arrays shaped like psiE / psiH holding, on
 *   entry, dLoss/d(psi) after the run and, on return, dLoss/d(psi) before it.
 * fdtdhip_adjoint_run: lamE, lamH (nx,ny,nz,3) hold dLoss/dE, dLoss/dH after the
 *   last of `nsteps` recorded steps on entry and before the first one on return;
 *   `work` is a scratch array of cells*6 elements.  Single-device grids only. */
int fdtdhip_plan_set_object_trace(fdtdhip_plan* plan, int id, void* trace, int64_t capacity);
int fdtdhip_plan_set_object_grad(fdtdhip_plan* plan, int id, void* grad);

/* FDTDHIP_OBJ_ANISOTROPIC_MODULATED: an anisotropic object (inv: (bx,by,bz,3,3)) whose
 * tensor entries are scaled, every E half step, by
 *     sigmoid(w[n][0]*sum(E^2) + w[n][1]*sum(H^2) + w[n][2]),   n = 3*row + column,
 * with the energies of the cell's own fields as they are when the object acts
 * (reference fdtd/objects.py:328-349, LearnableAnisotropicObject(is_substrate=True):
 * a 1x1 convolution from the two energies to nine channels, then a sigmoid).
 * `weights`: device (9,3), required before the first step and whenever the values
 * change.  Differentiable runs: `energy_trace` device (capacity, box, 2) filled by the
 * forward E half steps next to the curl trace (same capacity), read by the adjoint;
 * `grad_weights` device (box, 9, 3), to which the adjoint adds each cell's contribution to
 * dL/dweights (the caller sums over cells).  As in the reference no gradient flows
 * through the energies. */
int fdtdhip_plan_set_object_modifier(fdtdhip_plan* plan, int id, const void* weights,
                                     void* energy_trace, void* grad_weights);
int fdtdhip_plan_set_pml_adjoint(fdtdhip_plan* plan, int id, void* lam_psiE, void* lam_psiH);

/* Second adjoint arrays for whole reverse steps (no reference counterpart: torch autograd
 * walks the reference's tape one slice operation at a time, fdtd/grid.py:275-325).  With
 * lamE_alt / lamH_alt (device arrays shaped like lamE / lamH) and, for every PML,
 * lam_psiE_alt / lam_psiH_alt (shaped like its lam_psiE / lam_psiH), contents scratch, a plan
 * whose materials are proven constants, without periodic boundary and with additive point
 * sources only, undoes a step out of place, first -> second arrays: one marching pass over
 * the cells away from PML slabs and objects -- one read and one write of each adjoint field
 * -- and one pass per half step over the cells next to them, without the `work` array.
 * fdtdhip_adjoint_run then walks pairs of steps, first -> second arrays and back (an odd
 * first one in the two-pass form), and its results equal those without second arrays.
 * NULL withdraws them.  FDTDHIP_ADJ_MARCH=0 in the environment keeps the two-pass form. */
int fdtdhip_plan_set_adjoint_buffers(fdtdhip_plan* plan, void* lamE_alt, void* lamH_alt);
int fdtdhip_plan_set_pml_adjoint_buffers(fdtdhip_plan* plan, int id, void* lam_psiE_alt,
                                         void* lam_psiH_alt);
int fdtdhip_adjoint_run(fdtdhip_plan* plan, int64_t nsteps, void* lamE, void* lamH, void* work);

/* fdtdhip_adjoint_run for callers that cannot allocate device memory: tight serving /
 * training loops, harnesses that forbid hipMalloc, code that must not disturb a caching
 * allocator.  (Not for use while a HIP graph is being captured: the call waits for the
 * stream, see below.)  No counterpart in the reference.  Same inputs, outputs and results as fdtdhip_adjoint_run, without its
 * `work` argument: the work array, the second one the paired launches of small grids use,
 * and the PML / object descriptor tables live in one static device array of
 * FDTDHIP_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process
 * ends.  The call allocates and frees nothing (no hipMalloc / hipFree, no library routine
 * that allocates); it waits for the stream once per call while the descriptor tables are
 * copied.  Limits, refused with an error message otherwise:
 *   16384 + 2 * cells * 6 * sizeof(element) <= FDTDHIP_NOALLOC_SCRATCH_BYTES, the first of
 *   the two arrays rounded up to 256 bytes (349 354 cells in float64, 698 709 in float32),
 *   and at most 128 objects.
 * All calls on one GPU share that GPU's array.  Calls made one after the other from the host,
 * on any plans and streams, are ordered by the library: the sweep of each call waits, on its
 * own stream, for the event the call before it recorded behind its last kernel, and the work
 * arrays are cleared at the start of each call.  Calls made at the same time from two host
 * threads on one GPU race. */
#define FDTDHIP_NOALLOC_SCRATCH_BYTES (32u << 20)
int fdtdhip_adjoint_run_noalloc(fdtdhip_plan* plan, int64_t nsteps, void* lamE, void* lamH);

/* The same reverse sweep in pieces, for x-slabs (fdtdhip_plan_set_halo tells the plan on
 * which sides the slab continues).  The adjoint of one half step is
 *   fdtdhip_adjoint_cells  : reverse of the post pieces, then per cell the adjoint of
 *                            everything that followed the six neighbour differences ->
 *                            work[cell][6]; `step` is the row of the curl traces to use;
 *   fdtdhip_adjoint_gather : transpose of the difference stencil, added to the adjoint of
 *                            the driving field.  The differences across an x face between
 *                            slabs (reference fdtd/grid.py:46,48,71,73) make the gather reach
 *                            into the neighbour's `work`.  Only the adjoints of the two
 *                            differences along x cross the face -- entries 0 and 1 of the six
 *                            of a cell -- so `work_ghost` is a packed (ny, nz, 2) array: for
 *                            the E half step (phase_E = 1) work[plane 0][..., 0:2] of the rank
 *                            above, for the H half step work[last plane][..., 0:2] of the
 *                            rank below, exchanged by the caller between the two calls (the
 *                            size of the forward halo); NULL where the slab ends at a domain
 *                            edge.
 * A reverse step undoes the H half step first (phase_E = 0), then the E half step.
 * With FDTDHIP_PERIODIC_X_EXTERNAL in the plan (the copy between the first and the last rank
 * is the caller's, reference fdtd/boundaries.py:173-184) fdtdhip_adjoint_cells leaves the
 * post pieces to the caller, who runs, before it and in this order:
 *   fdtdhip_adjoint_post_stage(stage 1): sources and the boundaries registered after it;
 *   the transpose of the copy: the adjoint plane of the copy's destination (E half step:
 *   plane 0 of the first rank; H half step: the last plane of the last rank) is ADDED to
 *   the adjoint plane of its source on the other rank and then set to zero;
 *   fdtdhip_adjoint_post_stage(stage 0): the boundaries registered before it.
 * `work` belongs to the two calls of a half step: where the materials are proven constants
 * the library leaves out the entries of cells far from every PML, object and open slab
 * face and derives them inside the gather instead, so the caller may rely on the first and
 * last plane of an open slab only (the planes it exchanges). */
int fdtdhip_adjoint_cells(fdtdhip_plan* plan, int phase_E, int64_t step, void* lamE, void* lamH,
                          void* work);
int fdtdhip_adjoint_gather(fdtdhip_plan* plan, int phase_E, void* lamE, void* lamH, void* work,
                           const void* work_ghost);
int fdtdhip_adjoint_post_stage(fdtdhip_plan* plan, int phase_E, int64_t step, int stage, void* lamE,
                               void* lamH);

/* Whole reverse steps on an x-slab, in stages around the two plane exchanges of a step: the
 * mirror image of fdtdhip_step_stage (no reference counterpart).  With the second adjoint
 * arrays of fdtdhip_plan_set_adjoint_buffers / fdtdhip_plan_set_pml_adjoint_buffers set,
 * fdtdhip_plan_adjoint_whole_steps tells (1 / 0) whether the plan can take them.  One reverse
 * step from (lamE0, lamH0) to (lamE1, lamH1), `flip` = 1 when these are the first and the
 * second arrays (memory-variable adjoints first -> second) and 2 for the way back, `step` the
 * row of the curl traces:
 *   fdtdhip_adjoint_pack_plane(phase_E = 0)  the (ny, nz, 2) plane the rank ABOVE needs, from
 *            lamH0 of the last plane; the caller sends it and receives the rank below's;
 *   stage 0  H half step undone next to PMLs, objects and the high face, first half of the
 *            marching kernel over the other cells -- needs no plane;
 *   stage 1  H half step undone in plane 0: `ghost` = the plane received from the rank below;
 *   fdtdhip_adjoint_pack_plane(phase_E = 1)  the plane the rank BELOW needs, from the new
 *            lamE1 of plane 0; sent, and the rank above's received;
 *   stage 2  second half of the marching kernel, E half step undone away from the last plane;
 *   stage 3  E half step undone in the last plane: `ghost` = the plane received from above.
 * A stage without an open face on its side ignores `ghost` (NULL).  Results equal the
 * two-pass sweep of fdtdhip_adjoint_cells / fdtdhip_adjoint_gather.  `work`: cells*6 elements
 * of scratch.  Take steps in pairs so that the result ends in the first arrays. */
int fdtdhip_plan_adjoint_whole_steps(fdtdhip_plan* plan);
int fdtdhip_adjoint_pack_plane(fdtdhip_plan* plan, int phase_E, int64_t step, int flip, void* lamE,
                               void* lamH, void* out);
int fdtdhip_adjoint_step_stage(fdtdhip_plan* plan, int64_t step, int stage, int flip, void* lamE0,
                               void* lamH0, void* lamE1, void* lamH1, void* work, const void* ghost);

/* Optional: prove that the material arrays need not be read.  The reference fills
 * inverse_permittivity and inverse_permeability with one value (fdtd/grid.py:135-151)
 * and an object zeroes the inverse permittivity in its box (fdtd/objects.py:90), so
 * most grids hold a constant outside their objects.  This call reads both arrays once
 * on the plan's stream and waits for the answer; an array that is bit for bit
 * "constant outside the registered object boxes, +0 inside" (inverse permeability:
 * constant everywhere) is from then on passed to the half-step kernels as that
 * constant -- same results, a quarter less memory traffic.  Returns a bit mask
 * (1: inverse permittivity, 2: inverse permeability) or a negative error.
 * The finding lasts until fdtdhip_plan_set_fields names another array,
 * fdtdhip_plan_add_object or fdtdhip_plan_reset is called, or this function is
 * called again.  A caller that writes into a scanned array MUST call this function
 * again before the next half step; a caller that never calls it gets kernels that
 * read the arrays in full.  enable = 0 only forgets earlier findings (for arrays
 * that change every step).  FDTDHIP_MATERIAL_SCAN=0 in the environment turns the
 * scan into a no-op returning 0. */
int fdtdhip_plan_scan_materials(fdtdhip_plan* plan, int enable);

/* Energy densities sum(E^2) and sum(H^2) on the grid plane `index` normal to `axis`, the
 * quantity the reference's plots show (reference fdtd/visualization.py:119-146, which
 * squares and sums the whole grid to then keep one plane).  outE, outH: device arrays of
 * (ny,nz) for axis 0, (nz,nx) for axis 1 (the reference transposes that plane), (nx,ny)
 * for axis 2. */
int fdtdhip_plane_energy(fdtdhip_plan* plan, int axis, int index, void* outE, void* outH);

/* Kernel timing for benchmarks (the reference only has wall-clock timing around
 * grid.run, examples/old/simulation_benchmark.py:96-107).  While enabled, every
 * bulk kernel launch is bracketed by HIP events on the plan's stream.
 * fdtdhip_plan_get_timing synchronises those events and returns, accumulated
 * since the last call: total milliseconds and launch counts of the E and of the
 * H bulk kernel. */
int fdtdhip_plan_enable_timing(fdtdhip_plan* plan, int on);
int fdtdhip_plan_get_timing(fdtdhip_plan* plan, double* ms_E, int64_t* launches_E,
                            double* ms_H, int64_t* launches_H);
/* The same, by kind of launch: [0] H bulk kernels, [1] E bulk kernels, [2] whole-step kernels
 * (one launch advances E and H), [3] the half-step launches over the cells next to PMLs,
 * objects and open slab faces that accompany a whole-step kernel, [4] launches that advance E
 * and H by TWO steps (plans with nothing between their steps but additive source points).
 * fdtdhip_plan_get_timing reports [1] + [2] + [3] + [4] as the E time and [1] + [2] + [4] as
 * the E launches. */
int fdtdhip_plan_get_timing_parts(fdtdhip_plan* plan, double ms[5], int64_t launches[5]);

/* Stand-alone curls (reference fdtd/grid.py:29-76): out = curl(in), both device
 * (nx,ny,nz,3) arrays of `dtype`; enqueued on `stream`. */
int fdtdhip_curl_E(const void* E, void* out, int nx, int ny, int nz, int dtype, void* stream);
int fdtdhip_curl_H(const void* H, void* out, int nx, int ny, int nz, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FDTD_HIP_H */
