// Kernel micro-benchmark for libfdtd_hip: times the bulk half-step kernels through
// the C ABI on device-resident random data.  Usage:
//   kbench nx ny nz f32|f64 steps [pml_thickness]
// Kernel strategy is chosen with the library's FDTDHIP_BULK environment variable;
// KB_FUSED=1 provides second buffers so fdtdhip_run may take fused whole steps
// (FDTDHIP_FUSED=1 forces them on grids of any size).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/fdtd_hip.h"

#define CHECK(x)                                                        \
  do {                                                                  \
    if ((x) != hipSuccess) {                                            \
      fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__);      \
      exit(1);                                                          \
    }                                                                   \
  } while (0)
#define LIB(x)                                                          \
  do {                                                                  \
    if ((x) < 0) {                                                      \
      fprintf(stderr, "lib error: %s\n", fdtdhip_last_error());         \
      exit(1);                                                          \
    }                                                                   \
  } while (0)

template <typename T>
__global__ void fill(T* p, size_t n, unsigned seed, T lo, T hi) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    p[i] = (T)(lo + (hi - lo) * u);
  }
}

template <typename T>
void run(int nx, int ny, int nz, int steps, int t) {
  const size_t n = (size_t)nx * ny * nz * 3;
  T *E, *H, *ie, *im;
  CHECK(hipMalloc(&E, n * sizeof(T)));
  CHECK(hipMalloc(&H, n * sizeof(T)));
  CHECK(hipMalloc(&ie, n * sizeof(T)));
  CHECK(hipMalloc(&im, n * sizeof(T)));
  fill<T><<<4096, 256>>>(E, n, 1, (T)-1e-3, (T)1e-3);
  fill<T><<<4096, 256>>>(H, n, 2, (T)-1e-3, (T)1e-3);
  // KB_UNIFORM=1: homogeneous materials, which fdtdhip_plan_scan_materials turns into constants
  const bool uniform = getenv("KB_UNIFORM") && atoi(getenv("KB_UNIFORM"));
  fill<T><<<4096, 256>>>(ie, n, 3, uniform ? (T)1.0 : (T)0.5, (T)1.0);
  fill<T><<<4096, 256>>>(im, n, 4, uniform ? (T)1.0 : (T)0.5, (T)1.0);
  CHECK(hipDeviceSynchronize());
  fdtdhip_plan* plan;
  const int dtype = sizeof(T) == 4 ? FDTDHIP_F32 : FDTDHIP_F64;
  LIB(fdtdhip_plan_create(&plan, nx, ny, nz, dtype, 0.99 / 1.7320508075688772, 0));
  LIB(fdtdhip_plan_set_fields(plan, E, H, ie, im));
  const bool fused = getenv("KB_FUSED") && atoi(getenv("KB_FUSED"));
  if (fused) {
    T *E2, *H2;
    CHECK(hipMalloc(&E2, n * sizeof(T)));
    CHECK(hipMalloc(&H2, n * sizeof(T)));
    LIB(fdtdhip_plan_set_step_buffers(plan, E2, H2));
  }
  std::vector<void*> psis;
  if (t > 0) {
    std::vector<double> b(t, 0.9), c(t, -0.05);
    const int dims[3] = {nx, ny, nz};
    for (int axis = 0; axis < 3; ++axis)
      for (int side = 0; side < 2; ++side) {
        size_t slab = (size_t)t * nx * ny * nz / dims[axis] * 2;
        void *pe, *ph;
        CHECK(hipMalloc(&pe, slab * sizeof(T)));
        CHECK(hipMalloc(&ph, slab * sizeof(T)));
        CHECK(hipMemset(pe, 0, slab * sizeof(T)));
        CHECK(hipMemset(ph, 0, slab * sizeof(T)));
        psis.push_back(pe);
        psis.push_back(ph);
        const int id = fdtdhip_plan_add_pml(plan, axis, side, t, b.data(), c.data(), b.data(), c.data(), pe, ph);
        LIB(id);
        if (fused) {
          void* pa;
          CHECK(hipMalloc(&pa, slab * sizeof(T)));
          LIB(fdtdhip_plan_set_pml_step_buffer(plan, id, pa));
        }
      }
  }
  const int found = fdtdhip_plan_scan_materials(plan, uniform ? 1 : 0);
  LIB(found);
  LIB(fdtdhip_run(plan, 0, 4));
  CHECK(hipDeviceSynchronize());
  LIB(fdtdhip_plan_enable_timing(plan, 1));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, 0));
  LIB(fdtdhip_run(plan, 4, steps));
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float wall;
  CHECK(hipEventElapsedTime(&wall, e0, e1));
  double msE, msH;
  int64_t nE, nH;
  LIB(fdtdhip_plan_get_timing(plan, &msE, &nE, &msH, &nH));
  const double cells = (double)nx * ny * nz;
  // four arrays of three components; a material array found constant is not read
  const double bytes_half = cells * (found == 3 ? 9 : 12) * sizeof(T);
  const char* strat = getenv("FDTDHIP_BULK");
  if (nH == 0) { nH = 1; msH = 0; }
  printf("%-8s fused=%d const=%d %dx%dx%d %s pml=%d | E %.4f ms %.0f GB/s | H %.4f ms %.0f GB/s | step %.4f ms "
         "%.2f Gcells/s (%.0f GB/s algorithmic)\n",
         strat ? strat : "default", (int)fused, found, nx, ny, nz, sizeof(T) == 4 ? "f32" : "f64", t,
         msE / nE, bytes_half / (msE / nE) / 1e6, msH / nH, msH > 0 ? bytes_half / (msH / nH) / 1e6 : 0.0,
         wall / steps, cells * steps / wall / 1e6, 2 * bytes_half * steps / wall / 1e6);
  LIB(fdtdhip_plan_destroy(plan));
}

// Calibration: plain streaming kernels with the traffic mix of a half step
// (three arrays read, one written) and a 1:1 copy, float4 per lane.
__global__ void stream_3r1w(const float4* a, const float4* b, const float4* c, float4* o, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 x = a[i], y = b[i], z = c[i];
  float4 w = o[i];
  (void)w;
  o[i] = make_float4(x.x + y.x * z.x, x.y + y.y * z.y, x.z + y.z * z.z, x.w + y.w * z.w);
}
__global__ void stream_3r1w_pure(const float4* a, const float4* b, const float4* c, float4* o, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 x = a[i], y = b[i], z = c[i];
  o[i] = make_float4(x.x + y.x * z.x, x.y + y.y * z.y, x.z + y.z * z.z, x.w + y.w * z.w);
}
__global__ void stream_copy(const float4* a, float4* o, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i];
}
__global__ void stream_read(const float4* a, float4* o, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 x = a[i];
  if (x.x == 123.456f) o[0] = x;
}

template <int V>
__global__ void stream_3r1w_v(const float4* a, const float4* b, const float4* c, float4* o, size_t n) {
  const size_t i0 = ((size_t)blockIdx.x * blockDim.x) * V + threadIdx.x;
  float4 x[V], y[V], z[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const size_t i = i0 + (size_t)v * blockDim.x;
    if (i < n) { x[v] = a[i]; y[v] = b[i]; z[v] = c[i]; }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const size_t i = i0 + (size_t)v * blockDim.x;
    if (i < n) o[i] = make_float4(x[v].x + y[v].x * z[v].x, x[v].y + y[v].y * z[v].y,
                                  x[v].z + y[v].z * z[v].z, x[v].w + y[v].w * z[v].w);
  }
}

void calib(size_t stagger) {
  const size_t n4 = (size_t)512 * 512 * 512 * 3 / 4;  // one 512^3 float32 field, in float4s
  float4 *a, *b, *c, *o;
  char* pool;
  const size_t span = n4 * 16 + (64 << 20);
  CHECK(hipMalloc(&pool, 4 * span));
  a = (float4*)pool;
  b = (float4*)(pool + span + stagger);
  c = (float4*)(pool + 2 * span + 2 * stagger);
  o = (float4*)(pool + 3 * span + 3 * stagger);
  printf("calib: array offsets staggered by %zu bytes\n", stagger);
  fill<float><<<4096, 256>>>((float*)a, n4 * 4, 1, -1.f, 1.f);
  fill<float><<<4096, 256>>>((float*)b, n4 * 4, 2, -1.f, 1.f);
  fill<float><<<4096, 256>>>((float*)c, n4 * 4, 3, -1.f, 1.f);
  fill<float><<<4096, 256>>>((float*)o, n4 * 4, 4, -1.f, 1.f);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((n4 + 255) / 256);
  const int reps = 20;
  for (int mode = 0; mode < 6; ++mode) {
    float best = 1e9, sum = 0;
    for (int r = 0; r < reps + 2; ++r) {
      CHECK(hipEventRecord(e0, 0));
      if (mode == 0) stream_copy<<<grid, 256>>>(a, o, n4);
      else if (mode == 1) stream_read<<<grid, 256>>>(a, o, n4);
      else if (mode == 2) stream_3r1w_pure<<<grid, 256>>>(a, b, c, o, n4);
      else if (mode == 3) stream_3r1w<<<grid, 256>>>(a, b, c, o, n4);
      else if (mode == 4) stream_3r1w_v<2><<<(grid + 1) / 2, 256>>>(a, b, c, o, n4);
      else stream_3r1w_v<4><<<(grid + 3) / 4, 256>>>(a, b, c, o, n4);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)n4 * 16 * (mode == 0 ? 2 : mode == 1 ? 1 : 4);
    const char* names[6] = {"copy 1r1w", "read only", "3r1w", "3r1w (+read of output)",
                            "3r1w 2 float4/lane", "3r1w 4 float4/lane"};
    printf("calib %-24s avg %.4f ms %.0f GB/s | best %.4f ms %.0f GB/s\n", names[mode], sum / reps,
           bytes / (sum / reps) / 1e6, best, bytes / best / 1e6);
  }
}

// Infinity Cache probe: repeated read / write-then-read of buffers of growing size.
void mall_probe() {
  const size_t max_bytes = (size_t)1 << 31;
  float4 *a, *o;
  CHECK(hipMalloc(&a, max_bytes));
  CHECK(hipMalloc(&o, max_bytes));
  fill<float><<<4096, 256>>>((float*)a, max_bytes / 4, 1, -1.f, 1.f);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const size_t sizes_mb[] = {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048};
  for (size_t mb : sizes_mb) {
    const size_t n4 = mb * (1 << 20) / 16;
    const unsigned grid = (unsigned)((n4 + 255) / 256);
    float t_read = 0, t_copy = 0, t_wr = 0;
    for (int mode = 0; mode < 3; ++mode) {
      for (int r = 0; r < 12; ++r) {
        CHECK(hipEventRecord(e0, 0));
        if (mode == 0) stream_read<<<grid, 256>>>(a, o, n4);
        else if (mode == 1) stream_copy<<<grid, 256>>>(a, o, n4);
        else { stream_copy<<<grid, 256>>>(a, o, n4); stream_read<<<grid, 256>>>(o, a, n4); }
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) (mode == 0 ? t_read : mode == 1 ? t_copy : t_wr) += ms / 10;
      }
    }
    const double b = (double)n4 * 16;
    printf("mall %5zu MB | re-read %.0f GB/s | copy %.0f GB/s | copy then read-back of the copy %.0f GB/s (3 transfers)\n",
           mb, b / t_read / 1e6, 2 * b / t_copy / 1e6, 3 * b / t_wr / 1e6);
  }
}

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "mall")) { mall_probe(); return 0; }
  if (argc >= 2 && !strcmp(argv[1], "calib")) { calib(argc > 2 ? (size_t)atol(argv[2]) : 0); return 0; }
  if (argc < 6) {
    fprintf(stderr, "usage: kbench nx ny nz f32|f64 steps [pml]\n");
    return 2;
  }
  const int nx = atoi(argv[1]), ny = atoi(argv[2]), nz = atoi(argv[3]), steps = atoi(argv[5]);
  const int t = argc > 6 ? atoi(argv[6]) : 0;
  if (!strcmp(argv[4], "f32")) run<float>(nx, ny, nz, steps, t);
  else run<double>(nx, ny, nz, steps, t);
  return 0;
}
