// Experiment: one whole Yee step (E half step, then H half step) in a single pass over the
// grid, out of place, for float32 vacuum with constant materials -- 12 values moved per
// cell and step (E, H read; E, H written) instead of the 18 of two half-step passes.
//
// Form: a workgroup marches along x over a tile of TY owned rows x whole z rows.  A lane
// owns one z position of every row of the tile and keeps, per row, H and the new E of the
// previous plane in registers:
//   new E of plane i   needs H of plane i (own, row above, cell before) and of plane i-1
//   new H of plane i-1 needs new E of plane i-1 (own, row below, cell after) and of plane i
// The row below the last owned one is computed redundantly (its new E only); the cell
// after a wave's last lane belongs to the next wave of the workgroup and comes through
// LDS, one barrier per plane; the x range is cut in chunks (one redundant plane each).
//   kfuse n steps [ty] [lx]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../fdtd_amd/csrc/yee_march.hpp"   // the library's kernel, launched directly

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef float T;

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

// ---- reference: the two half steps as separate in-place passes ----------------------------
template <bool PHASE_E>
__global__ __launch_bounds__(256) void half_ref(T* F, const T* G, int nx, int ny, int nz, T sc, T m0,
                                                T m1, T m2) {
  const int64_t plane = (int64_t)ny * nz;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (tid >= plane * nx) return;
  const int i = (int)(tid / plane);
  const int64_t r = tid % plane;
  const int j = (int)(r / nz), k = (int)(r % nz);
  const int64_t sy = 3 * (int64_t)nz, sx = 3 * plane, c3 = tid * 3;
  const T gx = G[c3], gy = G[c3 + 1], gz = G[c3 + 2];
  T d_y_z = 0, d_y_x = 0, d_z_y = 0, d_z_x = 0, d_x_z = 0, d_x_y = 0;
  if (PHASE_E) {
    if (j > 0) { d_y_z = gz - G[c3 - sy + 2]; d_y_x = gx - G[c3 - sy]; }
    if (k > 0) { d_z_y = gy - G[c3 - 3 + 1]; d_z_x = gx - G[c3 - 3]; }
    if (i > 0) { d_x_z = gz - G[c3 - sx + 2]; d_x_y = gy - G[c3 - sx + 1]; }
    F[c3] += (sc * m0) * (d_y_z - d_z_y);
    F[c3 + 1] += (sc * m1) * (d_z_x - d_x_z);
    F[c3 + 2] += (sc * m2) * (d_x_y - d_y_x);
  } else {
    if (j < ny - 1) { d_y_z = G[c3 + sy + 2] - gz; d_y_x = G[c3 + sy] - gx; }
    if (k < nz - 1) { d_z_y = G[c3 + 3 + 1] - gy; d_z_x = G[c3 + 3] - gx; }
    if (i < nx - 1) { d_x_z = G[c3 + sx + 2] - gz; d_x_y = G[c3 + sx + 1] - gy; }
    F[c3] -= (sc * m0) * (d_y_z - d_z_y);
    F[c3 + 1] -= (sc * m1) * (d_z_x - d_x_z);
    F[c3 + 2] -= (sc * m2) * (d_x_y - d_y_x);
  }
}

struct ProtoArgs {
  const T* E0;
  const T* H0;
  T* E1;
  T* H1;
  int nx, ny, nz, lx;
  T sc;
  T me[3], mh[3];
};

__device__ __forceinline__ void ld3(const T* p, T& a, T& b, T& c, bool nt) {
  if (nt) {
    a = __builtin_nontemporal_load(p); b = __builtin_nontemporal_load(p + 1); c = __builtin_nontemporal_load(p + 2);
  } else {
    a = p[0]; b = p[1]; c = p[2];
  }
}

template <int TY, int NW>
__global__ __launch_bounds__(64 * NW) void proto_march(ProtoArgs a) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int k = w * 64 + lane;
  const bool kin = k < a.nz;
  const int j0 = blockIdx.x * TY;
  const int i0 = blockIdx.y * a.lx, i1 = min(i0 + a.lx, a.nx);
  const int64_t sy = 3 * (int64_t)a.nz, sx = sy * a.ny;
  // new E (components x, y) of each wave's first lane, for the last lane of the wave before it
  __shared__ T xch[2][NW + 1][TY][2];
  T Hp[TY + 1][3], Ep[TY + 1][3];
#pragma unroll
  for (int r = 0; r <= TY; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) { Hp[r][c] = 0; Ep[r][c] = 0; }
  if (i0 > 0 && kin) {
#pragma unroll
    for (int r = 0; r <= TY; ++r) {
      const int j = j0 + r;
      if (j < a.ny) ld3(a.H0 + (int64_t)(i0 - 1) * sx + (int64_t)j * sy + 3 * k, Hp[r][0], Hp[r][1], Hp[r][2], false);
    }
  }
  const bool hasz = k > 0, hasz1 = k < a.nz - 1;
  for (int i = i0; i <= i1; ++i) {
    T Hc[TY + 1][3], En[TY + 1][3];
#pragma unroll
    for (int r = 0; r <= TY; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) { Hc[r][c] = 0; En[r][c] = 0; }
    const bool plane = i < a.nx;
    if (plane && kin) {
      const int64_t pbase = (int64_t)i * sx + 3 * k;
      // all loads of the plane first
      T zx[TY + 1], zy[TY + 1], ux = 0, uz = 0;
#pragma unroll
      for (int r = 0; r <= TY; ++r) {
        const int j = j0 + r;
        if (j < a.ny) {
          const int64_t c3 = pbase + (int64_t)j * sy;
          ld3(a.H0 + c3, Hc[r][0], Hc[r][1], Hc[r][2], false);
          ld3(a.E0 + c3, En[r][0], En[r][1], En[r][2], true);
          const T* z = a.H0 + (hasz ? c3 - 3 : c3);
          zx[r] = z[0]; zy[r] = z[1];
        } else {
          Hc[r][0] = Hc[r][1] = Hc[r][2] = 0; En[r][0] = En[r][1] = En[r][2] = 0; zx[r] = zy[r] = 0;
        }
      }
      if (j0 > 0) {
        const T* u = a.H0 + pbase + (int64_t)(j0 - 1) * sy;
        ux = u[0]; uz = u[2];
      }
      const bool hasx = i > 0;
#pragma unroll
      for (int r = 0; r <= TY; ++r) {
        const int j = j0 + r;
        const bool hasy = j > 0;
        const T yx = r == 0 ? ux : Hc[r == 0 ? 0 : r - 1][0], yz = r == 0 ? uz : Hc[r == 0 ? 0 : r - 1][2];
        const T d_x_y = hasx ? Hc[r][1] - Hp[r][1] : T(0), d_x_z = hasx ? Hc[r][2] - Hp[r][2] : T(0);
        const T d_y_x = hasy ? Hc[r][0] - yx : T(0), d_y_z = hasy ? Hc[r][2] - yz : T(0);
        const T d_z_x = hasz ? Hc[r][0] - zx[r] : T(0), d_z_y = hasz ? Hc[r][1] - zy[r] : T(0);
        En[r][0] += (a.sc * a.me[0]) * (d_y_z - d_z_y);
        En[r][1] += (a.sc * a.me[1]) * (d_z_x - d_x_z);
        En[r][2] += (a.sc * a.me[2]) * (d_x_y - d_y_x);
        if (r < TY && j < a.ny && i < i1) {
          T* o = a.E1 + pbase + (int64_t)j * sy;
          __builtin_nontemporal_store(En[r][0], o);
          __builtin_nontemporal_store(En[r][1], o + 1);
          __builtin_nontemporal_store(En[r][2], o + 2);
        }
      }
    }
    if (i > i0) {
      // new H of plane i-1, owned rows
      const bool hasx1 = plane;   // plane i exists
#pragma unroll
      for (int r = 0; r < TY; ++r) {
        const int j = j0 + r;
        // new E of the cell after this one: the next lane's, or the next wave's first lane
        T nx_ = __shfl_down(Ep[r][0], 1), ny_ = __shfl_down(Ep[r][1], 1);
        if (lane == 63) { nx_ = xch[(i - 1) & 1][w + 1][r][0]; ny_ = xch[(i - 1) & 1][w + 1][r][1]; }
        if (kin && j < a.ny) {
          const bool hasy1 = j < a.ny - 1;
          const T d_x_y = hasx1 ? En[r][1] - Ep[r][1] : T(0), d_x_z = hasx1 ? En[r][2] - Ep[r][2] : T(0);
          const T d_y_x = hasy1 ? Ep[r + 1][0] - Ep[r][0] : T(0), d_y_z = hasy1 ? Ep[r + 1][2] - Ep[r][2] : T(0);
          const T d_z_x = hasz1 ? nx_ - Ep[r][0] : T(0), d_z_y = hasz1 ? ny_ - Ep[r][1] : T(0);
          T h0 = Hp[r][0], h1 = Hp[r][1], h2 = Hp[r][2];
          h0 -= (a.sc * a.mh[0]) * (d_y_z - d_z_y);
          h1 -= (a.sc * a.mh[1]) * (d_z_x - d_x_z);
          h2 -= (a.sc * a.mh[2]) * (d_x_y - d_y_x);
          T* o = a.H1 + (int64_t)(i - 1) * sx + (int64_t)j * sy + 3 * k;
          __builtin_nontemporal_store(h0, o);
          __builtin_nontemporal_store(h1, o + 1);
          __builtin_nontemporal_store(h2, o + 2);
        }
      }
    }
#pragma unroll
    for (int r = 0; r <= TY; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) { Hp[r][c] = Hc[r][c]; Ep[r][c] = En[r][c]; }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < TY; ++r) { xch[i & 1][w][r][0] = En[r][0]; xch[i & 1][w][r][1] = En[r][1]; }
    }
    __syncthreads();
  }
}

template <int TY>
void launch(const ProtoArgs& a) {
  const dim3 grid((a.ny + TY - 1) / TY, (a.nx + a.lx - 1) / a.lx);
  const int nw = (a.nz + 63) / 64;
  if (nw <= 4) proto_march<TY, 4><<<grid, 256>>>(a);
  else if (nw <= 8) proto_march<TY, 8><<<grid, 512>>>(a);
  else proto_march<TY, 16><<<grid, 1024>>>(a);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512, steps = argc > 2 ? atoi(argv[2]) : 20;
  const int ty = argc > 3 ? atoi(argv[3]) : 4, lx = argc > 4 ? atoi(argv[4]) : 32;
  const size_t cells = (size_t)n * n * n, ne = cells * 3;
  T *E0, *H0, *E1, *H1, *Er, *Hr;
  for (T** p : {&E0, &H0, &E1, &H1, &Er, &Hr}) CHECK(hipMalloc(p, ne * sizeof(T)));
  fill<<<4096, 256>>>(E0, ne, 1);
  fill<<<4096, 256>>>(H0, ne, 2);
  CHECK(hipMemcpy(Er, E0, ne * sizeof(T), hipMemcpyDeviceToDevice));
  CHECK(hipMemcpy(Hr, H0, ne * sizeof(T), hipMemcpyDeviceToDevice));
  const T sc = (T)(0.99 / 1.7320508075688772);
  const unsigned gb = (unsigned)((cells + 255) / 256);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  half_ref<true><<<gb, 256>>>(Er, Hr, n, n, n, sc, 1.0f, 0.9f, 0.8f);
  half_ref<false><<<gb, 256>>>(Hr, Er, n, n, n, sc, 0.7f, 1.0f, 0.95f);
  CHECK(hipDeviceSynchronize());
  T* href = (T*)malloc(ne * sizeof(T));
  T* hgot = (T*)malloc(ne * sizeof(T));
  ProtoArgs a;
  a.E0 = E0; a.H0 = H0; a.E1 = E1; a.H1 = H1;
  a.nx = a.ny = a.nz = n; a.lx = lx; a.sc = sc;
  a.me[0] = 1.0f; a.me[1] = 0.9f; a.me[2] = 0.8f;
  a.mh[0] = 0.7f; a.mh[1] = 1.0f; a.mh[2] = 0.95f;
  fdtdhip::MarchArgs<float> la;
  memset(&la, 0, sizeof(la));
  {
    fdtdhip::BulkArgs<float>& e = la.s.e;
    fdtdhip::BulkArgs<float>& h = la.s.h;
    e.F = E0; e.G = H0; h.F = H0; h.G = E0;
    for (fdtdhip::BulkArgs<float>* b : {&e, &h}) {
      b->nx = b->ny = b->nz = n; b->x0 = 0; b->x1 = n; b->sc = sc; b->mat_const = 1; b->trace_step = -1;
      b->rot_shift = 2;
    }
    e.bg[0] = 1.0f; e.bg[1] = 0.9f; e.bg[2] = 0.8f;
    h.bg[0] = 0.7f; h.bg[1] = 1.0f; h.bg[2] = 0.95f;
    la.s.E1 = E1; la.s.H1 = H1; la.s.nsrc = 0; la.s.detmap = nullptr;
    la.lx = lx; la.band = 0;
  }
  auto go = [&]() {
    if (ty == 0) {
      const dim3 g((n + 3) / 4, (n + lx - 1) / lx);
      fdtdhip::step_march<float, 4, false, false, true, 8><<<g, 64 * ((n + 63) / 64)>>>(la);
    }
    else if (ty == 2) launch<2>(a);
    else if (ty == 3) launch<3>(a);
    else if (ty == 5) launch<5>(a);
    else if (ty == 4) launch<4>(a);
    else if (ty == 6) launch<6>(a);
    else launch<8>(a);
  };
  go();
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  bool same = true;
  CHECK(hipMemcpy(href, Er, ne * sizeof(T), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hgot, E1, ne * sizeof(T), hipMemcpyDeviceToHost));
  const bool sameE = memcmp(href, hgot, ne * sizeof(T)) == 0;
  CHECK(hipMemcpy(href, Hr, ne * sizeof(T), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hgot, H1, ne * sizeof(T), hipMemcpyDeviceToHost));
  const bool sameH = memcmp(href, hgot, ne * sizeof(T)) == 0;
  same = sameE && sameH;
  if (!sameH) {
    size_t bad = 0, first = 0;
    for (size_t q = 0; q < ne; ++q)
      if (href[q] != hgot[q]) { if (!bad) first = q; ++bad; }
    printf("H differs in %zu elements, first at cell %zu comp %zu (plane %zu row %zu k %zu)\n", bad, first / 3,
           first % 3, first / 3 / ((size_t)n * n), first / 3 / n % n, first / 3 % n);
  }
  for (int r = 0; r < 3; ++r) go();
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < steps; ++r) go();
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= steps;
  printf("march ty=%d lx=%d %d^3 f32: %.4f ms per step, %.1f Gcells/s, %.0f GB/s of the 48 B/cell; E %s, H %s\n", ty, lx,
         n, ms, cells / ms / 1e6, cells * 48.0 / ms / 1e6, sameE ? "bitwise equal" : "DIFFERS",
         sameH ? "bitwise equal" : "DIFFERS");
  // the two half-step passes of the reference kernels, for scale
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < steps; ++r) {
    half_ref<true><<<gb, 256>>>(Er, Hr, n, n, n, sc, 1.0f, 0.9f, 0.8f);
    half_ref<false><<<gb, 256>>>(Hr, Er, n, n, n, sc, 0.7f, 1.0f, 0.95f);
  }
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("plain two-pass reference kernels: %.4f ms per step\n", ms / steps);
  return same ? 0 : 1;
}
