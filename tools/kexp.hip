// Experimental float64 E half-step kernels (vacuum, constant materials) timed against the
// library's form, to find out what the 24-byte cell costs the float64 path.
//   kexp n steps            (n^3 cells, float64)
// Variants:
//   cell    one lane per cell, three 8-byte components per lane (the library's mapping):
//           every load is a 16-byte + an 8-byte instruction at a 24-byte lane stride
//   scalar  one lane per component: lane l of a wave handles scalars l, 64+l, 128+l of the
//           wave's 64-cell row segment; every access is an 8-byte access, 512 contiguous
//           bytes per wave instruction for the advanced field
//   lds     one lane per cell for the arithmetic, but all global loads dense 8-byte
//           accesses staged through LDS
// Results are compared bit for bit with the plain reference kernel.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../fdtd_amd/csrc/yee_tiled.hpp"   // the library's kernels, launched directly

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef double T;

__global__ void fill(T* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    p[i] = (T)(-1e-3 + 2e-3 * ((double)(z >> 11) * (1.0 / 9007199254740992.0)));
  }
}

struct Args {
  T* F;
  const T* G;
  int nx, ny, nz;
  T sc;
  T bg[3];
};

// ---- reference: plain, conditional loads ------------------------------------------------
__global__ __launch_bounds__(256) void e_ref(Args a) {
  const int64_t plane = (int64_t)a.ny * a.nz;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (tid >= plane * a.nx) return;
  const int i = (int)(tid / plane);
  const int64_t r = tid % plane;
  const int j = (int)(r / a.nz), k = (int)(r % a.nz);
  const int64_t sy = 3 * (int64_t)a.nz, sx = 3 * plane, c3 = tid * 3;
  const T* G = a.G;
  const T hx = G[c3], hy = G[c3 + 1], hz = G[c3 + 2];
  T d_y_z = 0, d_y_x = 0, d_z_y = 0, d_z_x = 0, d_x_z = 0, d_x_y = 0;
  if (j > 0) { d_y_z = hz - G[c3 - sy + 2]; d_y_x = hx - G[c3 - sy]; }
  if (k > 0) { d_z_y = hy - G[c3 - 3 + 1]; d_z_x = hx - G[c3 - 3]; }
  if (i > 0) { d_x_z = hz - G[c3 - sx + 2]; d_x_y = hy - G[c3 - sx + 1]; }
  const T cu0 = d_y_z - d_z_y, cu1 = d_z_x - d_x_z, cu2 = d_x_y - d_y_x;
  a.F[c3] += (a.sc * a.bg[0]) * cu0;
  a.F[c3 + 1] += (a.sc * a.bg[1]) * cu1;
  a.F[c3 + 2] += (a.sc * a.bg[2]) * cu2;
}

__device__ __forceinline__ bool lane_cell(int ny, int nz, int& i, int& j, int& k) {
  k = ((blockIdx.x + (blockIdx.y >> 2)) % gridDim.x) * 64 + (threadIdx.x & 63);
  j = blockIdx.y * 4 + (threadIdx.x >> 6);
  i = blockIdx.z;
  return k < nz && j < ny;
}

// ---- cell per lane, unconditional loads, store through LDS (the library's form) -----------
__global__ __launch_bounds__(256) void e_cell(Args a) {
  int i, j, k;
  if (!lane_cell(a.ny, a.nz, i, j, k)) return;
  const int64_t sy = 3 * (int64_t)a.nz, sx = sy * a.ny;
  const int64_t c3 = (int64_t)i * sx + (int64_t)j * sy + 3 * k;
  const T* g = a.G + c3;
  const bool hasx = i > 0, hasy = j > 0, hasz = k > 0;
  const T* gx = hasx ? g - sx : g;
  const T* gy = hasy ? g - sy : g;
  const T* gz = hasz ? g - 3 : g;
  const T hx = g[0], hy = g[1], hz = g[2];
  const T xy = gx[1], xz = gx[2];
  const T yx = gy[0], yz = gy[2];
  const T zx = gz[0], zy = gz[1];
  T e0 = a.F[c3], e1 = a.F[c3 + 1], e2 = a.F[c3 + 2];
  const T d_x_y = hasx ? hy - xy : T(0), d_x_z = hasx ? hz - xz : T(0);
  const T d_y_x = hasy ? hx - yx : T(0), d_y_z = hasy ? hz - yz : T(0);
  const T d_z_x = hasz ? hx - zx : T(0), d_z_y = hasz ? hy - zy : T(0);
  e0 += (a.sc * a.bg[0]) * (d_y_z - d_z_y);
  e1 += (a.sc * a.bg[1]) * (d_z_x - d_x_z);
  e2 += (a.sc * a.bg[2]) * (d_x_y - d_y_x);
  __shared__ T stage[4 * 192];
  const int lane = threadIdx.x & 63;
  volatile T* l = stage + (threadIdx.x >> 6) * 192;
  if (__builtin_amdgcn_read_exec() != ~0ull) {
    a.F[c3] = e0; a.F[c3 + 1] = e1; a.F[c3 + 2] = e2;
    return;
  }
  l[lane * 3] = e0; l[lane * 3 + 1] = e1; l[lane * 3 + 2] = e2;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  T* out = a.F + c3 - lane * 3;
  __builtin_nontemporal_store(l[lane * 2], out + lane * 2);
  __builtin_nontemporal_store(l[lane * 2 + 1], out + lane * 2 + 1);
  if (lane < 32) {
    __builtin_nontemporal_store(l[128 + lane * 2], out + 128 + lane * 2);
    __builtin_nontemporal_store(l[128 + lane * 2 + 1], out + 128 + lane * 2 + 1);
  }
}

// ---- one lane per scalar ------------------------------------------------------------------
// NT: non-temporal access to the advanced field (read once, written once)
template <bool NT>
__global__ __launch_bounds__(256) void e_scalar(Args a) {
  const int lane = threadIdx.x & 63;
  const int k0 = ((blockIdx.x + (blockIdx.y >> 2)) % gridDim.x) * 64;   // first cell of the segment
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int i = blockIdx.z;
  if (j >= a.ny) return;
  const int64_t sy = 3 * (int64_t)a.nz, sx = sy * a.ny;
  const int64_t seg = (int64_t)i * sx + (int64_t)j * sy + 3 * (int64_t)k0;   // first scalar of the segment
  const int nscal = 3 * min(64, a.nz - k0);
  T ga[3], gan[3], gb[3], gbn[3], f[3];
  bool ma[3], mb[3], live[3];
  int cc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const int s = g * 64 + lane;           // scalar within the segment
    live[g] = s < nscal;
    const int cell = s / 3, c = s - cell * 3;
    cc[g] = c;
    const int ca = c == 0 ? 2 : c - 1, p = c == 2 ? 0 : c + 1;      // a = (c+2)%3, p = (c+1)%3
    const int cb = p;                                               // b = (c+1)%3, q = (c+2)%3
    const bool hz_ = k0 + cell > 0;
    // c = 0: p = y, q = z;  c = 1: p = z, q = x;  c = 2: p = x, q = y
    ma[g] = c == 0 ? j > 0 : (c == 1 ? hz_ : i > 0);
    mb[g] = c == 0 ? hz_ : (c == 1 ? i > 0 : j > 0);
    const int64_t sp = c == 0 ? sy : (c == 1 ? 3 : sx);
    const int64_t sq = c == 0 ? 3 : (c == 1 ? sx : sy);
    const int64_t base = seg + 3 * cell;
    const T* G = a.G;
    if (live[g]) {
      ga[g] = G[base + ca];
      gb[g] = G[base + cb];
      gan[g] = G[ma[g] ? base + ca - sp : base + ca];
      gbn[g] = G[mb[g] ? base + cb - sq : base + cb];
      f[g] = NT ? __builtin_nontemporal_load(a.F + seg + s) : a.F[seg + s];
    }
  }
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    if (!live[g]) continue;
    const T da = ma[g] ? ga[g] - gan[g] : T(0);
    const T db = mb[g] ? gb[g] - gbn[g] : T(0);
    const T m = cc[g] == 0 ? a.bg[0] : (cc[g] == 1 ? a.bg[1] : a.bg[2]);
    const T v = f[g] + (a.sc * m) * (da - db);
    if (NT) __builtin_nontemporal_store(v, a.F + seg + g * 64 + lane);
    else a.F[seg + g * 64 + lane] = v;
  }
}

// ---- cell per lane, global accesses dense 8-byte accesses staged through LDS ---------------
// XLDS / YLDS: the x / y neighbour segment goes through LDS too (dense loads); otherwise the
// lane loads the two components it needs of its neighbour cell directly (x: components 1, 2
// are adjacent, one 16-byte load; y: components 0 and 2, two 8-byte loads).
template <bool XLDS, bool YLDS>
__global__ __launch_bounds__(256) void e_lds(Args a) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k0 = ((blockIdx.x + (blockIdx.y >> 2)) % gridDim.x) * 64;
  const int j = blockIdx.y * 4 + w;
  const int i = blockIdx.z;
  if (j >= a.ny) return;
  const int ncell = min(64, a.nz - k0);
  const int nscal = 3 * ncell;
  const int64_t sy = 3 * (int64_t)a.nz, sx = sy * a.ny;
  const int64_t seg = (int64_t)i * sx + (int64_t)j * sy + 3 * (int64_t)k0;
  const bool hasx = i > 0, hasy = j > 0;
  // per wave: own segment preceded by one cell (the z neighbour of its first cell), the
  // advanced field, and the neighbour segments that are staged
  constexpr int PER_WAVE = 4 + 192 * (2 + (XLDS ? 1 : 0) + (YLDS ? 1 : 0));
  __shared__ T lds[4][PER_WAVE];
  T* L = lds[w];
  T* own = L + 3;            // own[-3..-1]: the cell before the segment
  T* ff = L + 3 + 192;
  T* nx_ = L + 3 + 384;
  T* ny_ = L + 3 + 384 + (XLDS ? 192 : 0);
  const T* G = a.G;
  const int64_t offx = hasx ? -sx : 0, offy = hasy ? -sy : 0;
  const bool active = lane < ncell;
  T v0[3], v1[3], v2[3], v3[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const int s = g * 64 + lane;
    if (s < nscal) {
      v0[g] = G[seg + s];
      if (XLDS) v1[g] = G[seg + offx + s];
      if (YLDS) v2[g] = G[seg + offy + s];
      v3[g] = __builtin_nontemporal_load(a.F + seg + s);
    }
  }
  T before = 0;
  if (lane < 3 && k0 > 0) before = G[seg - 3 + lane];
  T xy = 0, xz = 0, yx = 0, yz = 0;
  if (!XLDS && active) { xy = G[seg + offx + 3 * lane + 1]; xz = G[seg + offx + 3 * lane + 2]; }
  if (!YLDS && active) { yx = G[seg + offy + 3 * lane]; yz = G[seg + offy + 3 * lane + 2]; }
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const int s = g * 64 + lane;
    if (s < nscal) {
      own[s] = v0[g]; ff[s] = v3[g];
      if (XLDS) nx_[s] = v1[g];
      if (YLDS) ny_[s] = v2[g];
    }
  }
  if (lane < 3) L[lane] = before;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  T e0 = 0, e1 = 0, e2 = 0;
  if (active) {
    const bool hasz = k0 + lane > 0;
    const T hx = own[3 * lane], hy = own[3 * lane + 1], hz = own[3 * lane + 2];
    if (XLDS) { xy = nx_[3 * lane + 1]; xz = nx_[3 * lane + 2]; }
    if (YLDS) { yx = ny_[3 * lane]; yz = ny_[3 * lane + 2]; }
    const T zx = own[3 * lane - 3], zy = own[3 * lane - 2];
    e0 = ff[3 * lane]; e1 = ff[3 * lane + 1]; e2 = ff[3 * lane + 2];
    const T d_x_y = hasx ? hy - xy : T(0), d_x_z = hasx ? hz - xz : T(0);
    const T d_y_x = hasy ? hx - yx : T(0), d_y_z = hasy ? hz - yz : T(0);
    const T d_z_x = hasz ? hx - zx : T(0), d_z_y = hasz ? hy - zy : T(0);
    e0 += (a.sc * a.bg[0]) * (d_y_z - d_z_y);
    e1 += (a.sc * a.bg[1]) * (d_z_x - d_x_z);
    e2 += (a.sc * a.bg[2]) * (d_x_y - d_y_x);
  }
  __builtin_amdgcn_wave_barrier();
  if (active) { ff[3 * lane] = e0; ff[3 * lane + 1] = e1; ff[3 * lane + 2] = e2; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const int s = g * 64 + lane;
    if (s < nscal) __builtin_nontemporal_store(ff[s], a.F + seg + s);
  }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512, steps = argc > 2 ? atoi(argv[2]) : 20;
  const size_t cells = (size_t)n * n * n, ne = cells * 3;
  T *F0, *G, *F, *Fref;
  CHECK(hipMalloc(&F0, ne * sizeof(T)));
  CHECK(hipMalloc(&G, ne * sizeof(T)));
  CHECK(hipMalloc(&F, ne * sizeof(T)));
  CHECK(hipMalloc(&Fref, ne * sizeof(T)));
  fill<<<4096, 256>>>(F0, ne, 1);
  fill<<<4096, 256>>>(G, ne, 2);
  Args a;
  a.G = G; a.nx = a.ny = a.nz = n; a.sc = 0.99 / 1.7320508075688772;
  a.bg[0] = 1.0; a.bg[1] = 0.9; a.bg[2] = 0.8;
  CHECK(hipMemcpy(Fref, F0, ne * sizeof(T), hipMemcpyDeviceToDevice));
  a.F = Fref;
  e_ref<<<(unsigned)((cells + 255) / 256), 256>>>(a);
  CHECK(hipDeviceSynchronize());
  T* href = (T*)malloc(ne * sizeof(T));
  T* hgot = (T*)malloc(ne * sizeof(T));
  CHECK(hipMemcpy(href, Fref, ne * sizeof(T), hipMemcpyDeviceToHost));
  const dim3 rows((n + 63) / 64, (n + 3) / 4, n);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  fdtdhip::BulkArgs<double> la;
  memset(&la, 0, sizeof(la));
  la.G = G; la.mat = nullptr; la.ghost = nullptr;
  la.nx = la.ny = la.nz = n; la.x0 = 0; la.x1 = n;
  la.sc = a.sc; la.mat_const = 1;
  la.bg[0] = a.bg[0]; la.bg[1] = a.bg[1]; la.bg[2] = a.bg[2];
  la.rot_shift = 2; la.trace_step = -1;
  const char* names[] = {"cell", "scalar", "scalar+nt", "lds xy", "lds y", "lds x", "lds -", "library E"};
  // order of the variants: argv[3] as a string of digits (default: all in turn)
  const char* order = argc > 3 ? argv[3] : "01234567";
  for (const char* o = order; *o; ++o) {
    const int v = *o - '0';
    if (v < 0 || v > 7) continue;
    la.F = F;
    CHECK(hipMemcpy(F, F0, ne * sizeof(T), hipMemcpyDeviceToDevice));
    a.F = F;
    auto launch = [&]() {
      if (v == 0) e_cell<<<rows, 256>>>(a);
      else if (v == 1) e_scalar<false><<<rows, 256>>>(a);
      else if (v == 2) e_scalar<true><<<rows, 256>>>(a);
      else if (v == 3) e_lds<true, true><<<rows, 256>>>(a);
      else if (v == 4) e_lds<false, true><<<rows, 256>>>(a);
      else if (v == 5) e_lds<true, false><<<rows, 256>>>(a);
      else if (v == 6) e_lds<false, false><<<rows, 256>>>(a);
      else fdtdhip::bulk_E_rows<double, false, false, false, false, true><<<rows, 256>>>(la);
    };
    launch();
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hgot, F, ne * sizeof(T), hipMemcpyDeviceToHost));
    const bool same = memcmp(hgot, href, ne * sizeof(T)) == 0;
    for (int r = 0; r < 3; ++r) launch();
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < steps; ++r) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= steps;
    printf("%-10s %d^3 f64: %.4f ms per launch, %.0f GB/s of the 72 B/cell, bitwise equal to the reference: %s\n",
           names[v], n, ms, cells * 72.0 / ms / 1e6, same ? "yes" : "NO");
  }
  return 0;
}
