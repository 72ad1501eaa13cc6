// Plan data structures shared by the kernels and the C ABI of libfdtd_hip.
//
// Field layout everywhere: C-contiguous (nx, ny, nz, 3) -- cell major, the three
// vector components of one cell adjacent -- exactly the arrays the reference
// keeps on its Grid (reference fdtd/grid.py:135-151).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fdtdhip {

constexpr int MAX_PML = 6;  // one per face: the reference refuses a second boundary on a face
                            // (reference fdtd/boundaries.py:296-343)

// One PML face as the kernels see it, for one half step (E or H).
template <typename T>
struct PmlK {
  T* psi;       // (2, slab cells): [0] -> component u=(axis+1)%3, [1] -> component v=(axis+2)%3
  T* psi_out;   // where the advanced memory variables go: `psi` itself, or the second
                // buffer of a fused full step (neighbour lanes still read the old values)
  const T* b;   // (t) decay of the memory variable along the normal axis
  const T* c;   // (t) gain of the field difference along the normal axis
  int axis;     // normal axis
  int lo;       // first grid index of the slab along `axis`
  int t;        // thickness
  int fused;    // 1: the field correction is applied inside the bulk kernel;
                // 0: a periodic boundary precedes this PML in the boundary list, so the
                //    correction is applied by a later slab kernel
  int sy, sz;   // slab extents along y and z (x extent not needed for indexing)
};

// One object box (reference fdtd/objects.py).
template <typename T>
struct ObjK {
  const T* inv;     // (box,3) or (box,3,3)
  const T* absorb;  // (box,3) or null
  T* trace;         // differentiable runs: (steps, box, 3) curl of the driving field inside the
                    // box, written by the forward E half step, read by the adjoint; or null
  T* grad;          // adjoint: accumulated gradient of the loss w.r.t. `inv`, same shape; or null
  // kind 3 (anisotropic, tensor modulated by the local field energies):
  const T* modw;    // (9, 3): per tensor entry the weights of sum(E^2), sum(H^2) and the bias
  T* etrace;        // differentiable runs: (steps, box, 2) the two energies per step; or null
  T* gmod;          // adjoint: (box, 9, 3) per-cell gradient contributions to `modw`; or null
  int kind;
  int x0, x1, y0, y1, z0, z1;
};

// Arguments of the fused bulk kernels.  `F` is the field being advanced, `G` the
// field whose curl drives it, `mat` the matching inverse material array.
template <typename T>
struct BulkArgs {
  T* F;
  const T* G;
  const T* mat;
  const T* ghost;  // x-slabs: the two components of G that cross an x face (y, z), taken from
                   // the neighbour slab's adjacent plane, (ny,nz,2); null at a true domain edge
  int nx, ny, nz;
  int x0, x1;      // planes to process
  T sc;            // Courant number
  int npml;
  int nobj;
  const ObjK<T>* obj;
  // One object's trace and gradient arrays may travel with the kernel arguments instead of in
  // the table (load_obj): a tape that records one step per node binds new ones for every node,
  // and a table upload per node costs more than the step's launch.  ovr_obj: its index, or -1.
  int ovr_obj;
  T* ovr_trace;
  T* ovr_grad;
  int64_t trace_step;  // row of the objects' curl traces to write (forward) / read (adjoint); -1: none
  const PmlK<T>* pml;  // device table of npml descriptors (read with scalar loads, one at a time)
  int mat_const;       // 1: `mat` was scanned and holds `bg` in every cell outside the object
                       // boxes and zero inside them; the kernels then do not read it
  T bg[3];
  int rot_shift;       // row kernel: the z segment of a workgroup advances by one every
                       // 2^rot_shift block rows (see lane_cell)
  int hull[6];         // x0,x1,y0,y1,z0,z1 of the box that contains every object: a wavefront
                       // with no cell inside it skips the object table
  int core[6];         // x0,x1,y0,y1,z0,z1 of the box no PML slab reaches: a wavefront whose
                       // cells all lie inside it skips the PML table altogether
  // Normal axis and index range [lo, hi) of each PML face, in table order: a wavefront
  // near some slab tests every face against these (already in scalar registers with the
  // other arguments) and reads from the table only the descriptors of faces it touches.
  int face_axis[MAX_PML], face_lo[MAX_PML], face_hi[MAX_PML];
};

// Arguments of the adjoint (reverse-time) half-step kernels.  For the adjoint of
// the E half step `lamF` is the adjoint of E and `lamG` the adjoint of H.
template <typename T>
struct AdjArgs {
  T* lamF;         // in/out: adjoint of the field the forward half step advanced
  T* lamG;         // accumulated: adjoint of the driving field
  const T* mat;
  T* work;         // (cells, 6) adjoints of the six neighbour differences
  int nx, ny, nz;
  T sc;
  int npml;
  int nobj;
  const ObjK<T>* obj;
  int ovr_obj;         // as in BulkArgs
  T* ovr_trace;
  T* ovr_grad;
  int64_t trace_step;
  const PmlK<T>* pml;  // device table; psi -> adjoint of the memory variables
  // x-slabs: the slab continues below plane 0 / above plane nx-1 on a neighbour rank, so the
  // x differences at that face exist (they read the neighbour's plane in the forward pass)
  int open_lo, open_hi;
  const T* wghost;     // pass 2: across the face the gather reaches over, the two difference
                       // adjoints along x (WX_Y, WX_Z) of the neighbour's plane, (ny,nz,2); or null
  int mat_const;       // as in BulkArgs: `mat` is bg outside the object boxes, zero inside
  T bg[3];
  int hull[6];         // box containing every object
  int core[6];         // box no PML slab reaches
  int rot_shift;       // row mapping: z-segment rotation period (as in BulkArgs)
  int direct;          // cells far from every PML and object skip pass 1, and pass 2 derives
                       // their difference adjoints from lamF itself
  // normal axis and index range [lo, hi) of each PML face, in table order (as in BulkArgs): a
  // wavefront reads the descriptor of a face only if one of its cells lies in that face
  int face_axis[MAX_PML], face_lo[MAX_PML], face_hi[MAX_PML];
};

// Descriptor `o` of the object table, with the arrays that travel with the arguments in place.
template <typename T, typename Args>
__device__ __forceinline__ ObjK<T> load_obj(const Args& a, int o) {
  ObjK<T> O = a.obj[o];
  if (o == a.ovr_obj) {
    O.trace = a.ovr_trace;
    O.grad = a.ovr_grad;
  }
  return O;
}

}  // namespace fdtdhip
