// Does a persistent wave that prefetches its next tile overlap memory and arithmetic better than one wave per tile?
// The access pattern of a ONE-element chain with the fused read-out (C5: 8 fp64 streams + 1 byte read, 8 + 3 fp64 streams
// + 1 byte written per ray), with K dependent fp64 FMAs per ray between the loads and the stores (C5: ~650 VALU
// instructions per wave), in two forms:
//   oneshot   one workgroup per 256 rays: load, K FMAs, store (the shipped structure); W workgroups resident per CU
//   prefetch  256 x W workgroups, each looping over tiles: the loads of tile t + 1 are issued before the arithmetic of tile t
// Build: hipcc -O3 --offload-arch=gfx950 tools/overlap_probe.hip -o tools/_build/overlap_probe ; run: overlap_probe [n]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kB = 256;

struct Tile { double v[8]; uint8_t a; };   // (the byte is converted where it is used: a conversion at the load would wait for it)

__device__ __forceinline__ void load_tile(Tile& t, const double* __restrict__ in, const uint8_t* __restrict__ ain, int64_t n, int64_t i) {
#pragma unroll
  for (int f = 0; f < 8; ++f) t.v[f] = __builtin_nontemporal_load(in + f * n + i);
  t.a = ain[i];
}

template <int K>
__device__ __forceinline__ void work(Tile& t, const double c0, const double c1) {
  // four dependent chains of K / 4 FMAs each, mixing all eight inputs (nothing the compiler can fold)
  double a0 = t.v[0] + (double)t.a, a1 = t.v[1], a2 = t.v[2], a3 = t.v[3];
#pragma unroll 8
  for (int k = 0; k < K / 4; ++k) {
    a0 = fma(a0, c0, t.v[4]);
    a1 = fma(a1, c1, t.v[5]);
    a2 = fma(a2, c0, t.v[6]);
    a3 = fma(a3, c1, t.v[7]);
  }
  t.v[0] = a0; t.v[1] = a1; t.v[2] = a2; t.v[3] = a3;
}

__device__ __forceinline__ void store_tile(const Tile& t, double* __restrict__ out, uint8_t* __restrict__ aout, double* __restrict__ ro,
                                           int64_t n, int64_t i) {
#pragma unroll
  for (int f = 0; f < 8; ++f) __builtin_nontemporal_store(t.v[f], out + f * n + i);
  __builtin_nontemporal_store((uint8_t)1, aout + i);
#pragma unroll
  for (int f = 0; f < 3; ++f) __builtin_nontemporal_store(t.v[f] + 9.0, ro + (int64_t)f * n + i);
}

template <int K>
__global__ __launch_bounds__(kB) void k_oneshot(const double* __restrict__ in, const uint8_t* __restrict__ ain, double* __restrict__ out,
                                                uint8_t* __restrict__ aout, double* __restrict__ ro, int64_t n, double c0, double c1) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  Tile t;
  load_tile(t, in, ain, n, i);
  work<K>(t, c0, c1);
  store_tile(t, out, aout, ro, n, i);
}

template <int K>
__global__ __launch_bounds__(kB) void k_prefetch(const double* __restrict__ in, const uint8_t* __restrict__ ain, double* __restrict__ out,
                                                 uint8_t* __restrict__ aout, double* __restrict__ ro, int64_t n, double c0, double c1) {
  const int64_t stride = (int64_t)gridDim.x * kB;
  int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;      // n is a multiple of 256: a workgroup's tiles are whole
  Tile nxt;
  load_tile(nxt, in, ain, n, i);
  while (true) {
    Tile cur = nxt;
    const int64_t j = i + stride;
    const bool more = j < n;                               // uniform per workgroup
    if (more) load_tile(nxt, in, ain, n, j);
    work<K>(cur, c0, c1);
    store_tile(cur, out, aout, ro, n, i);
    if (!more) break;
    i = j;
  }
}

// the same loop without the prefetch (loads of tile t + 1 issued after the stores of tile t): what the loop alone does
template <int K>
__global__ __launch_bounds__(kB) void k_loop(const double* __restrict__ in, const uint8_t* __restrict__ ain, double* __restrict__ out,
                                             uint8_t* __restrict__ aout, double* __restrict__ ro, int64_t n, double c0, double c1) {
  const int64_t stride = (int64_t)gridDim.x * kB;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < n; i += stride) {
    Tile t;
    load_tile(t, in, ain, n, i);
    work<K>(t, c0, c1);
    store_tile(t, out, aout, ro, n, i);
  }
}

template <class F>
static float timeit(F launch, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

template <int K>
static int run(int64_t n, double* in, uint8_t* ain, double* out, uint8_t* aout, double* ro) {
  const int64_t nb = n / kB;
  const double bytes = (65.0 + 89.0) * n;
  const int reps = 60;
  const double c0 = 0.999999, c1 = 1.000001;
  auto line = [&](const char* name, int w, float ms) {
    printf("K=%4d  %-10s %d WG/CU  %8.4f ms  %6.3f TB/s  %6.1f GFMA/s\n", K, name, w, ms, bytes / ms * 1e-9, (double)K * n / ms * 1e-6);
    fflush(stdout);
  };
  const int ws[] = {8, 5, 4, 3, 2};
  for (int w : ws) {
    const int lds = w == 8 ? 0 : 160 * 1024 / w - 1024;      // dynamic LDS as the occupancy limiter
    line("oneshot", w, timeit([&] { k_oneshot<K><<<nb, kB, lds>>>(in, ain, out, aout, ro, n, c0, c1); }, reps));
  }
  for (int w : ws) {
    const int lds = w == 8 ? 0 : 160 * 1024 / w - 1024;
    line("loop", w, timeit([&] { k_loop<K><<<256 * w, kB, lds>>>(in, ain, out, aout, ro, n, c0, c1); }, reps));
    line("prefetch", w, timeit([&] { k_prefetch<K><<<256 * w, kB, lds>>>(in, ain, out, aout, ro, n, c0, c1); }, reps));
  }
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  return 0;
}

int main(int argc, char** argv) {
  int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
  if (n <= 0 || n > 100000000) { printf("n out of range\n"); return 2; }
  n = (n + kB - 1) / kB * kB;
  double *in, *out, *ro; uint8_t *ain, *aout;
  CK(hipMalloc(&in, 8 * n * 8)); CK(hipMalloc(&out, 8 * n * 8)); CK(hipMalloc(&ro, 3 * n * 8)); CK(hipMalloc(&ain, n)); CK(hipMalloc(&aout, n));
  CK(hipMemset(in, 0, 8 * n * 8)); CK(hipMemset(ain, 1, n));
  for (int pass = 0; pass < 2; ++pass) {       // the first pass also brings the clocks up; read the second
    printf("pass %d\n", pass);
    if (run<0>(n, in, ain, out, aout, ro)) return 1;
    if (run<320>(n, in, ain, out, aout, ro)) return 1;
    if (run<640>(n, in, ain, out, aout, ro)) return 1;
    if (run<1280>(n, in, ain, out, aout, ro)) return 1;
  }
  hipFree(in); hipFree(out); hipFree(ain); hipFree(aout); hipFree(ro);
  return 0;
}
