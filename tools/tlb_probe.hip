// Why does the same launch take up to 25 % longer in one allocation than in another of the same size (tools/pitch_probe.py)?
// Hypothesis: address translation.  A workgroup of the fused chain kernel writes 8 E + 1 streams whose rows lie n x 8 B
// (80-100 MB) apart: that many distinct pages in flight per workgroup, whatever the page size below ~100 MB -- and how large
// the driver's translation fragments are differs from allocation to allocation.  Test: the tracing pattern (7 + 1 streams
// read, E x (8 + 1) written per ray) into B buffers of the same size, in two layouts of the SAME bytes:
//   rows     the shipped layout: row r of the output block at r * n * 8 B
//   blocked  rows interleaved in tiles of T rays: element (r, i) at ((i / T) * R + r) * T + i % T -- all rows of a tile
//            within R * T * 8 B (16 MB for R = 64, T = 32768), i.e. a handful of 2-MB pages instead of R
// If the spread between buffers disappears in the blocked layout, translation is what the lottery is about.
// Build: hipcc -O3 --offload-arch=gfx950 tools/tlb_probe.hip -o tools/_build/tlb_probe ; run: tlb_probe [n] [buffers]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kB = 256;

template <int E, int T>   // T = 0: rows; else rays per tile (a multiple of 256)
__global__ __launch_bounds__(kB) void k_pattern(const double* __restrict__ in, const uint8_t* __restrict__ ain, double* __restrict__ out,
                                                uint8_t* __restrict__ aout, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  double v[8];
#pragma unroll
  for (int f = 0; f < 7; ++f) v[f] = __builtin_nontemporal_load(in + f * n + i);
  v[7] = (double)ain[i];
  constexpr int R = 8 * E;
#pragma unroll
  for (int e = 0; e < E; ++e) {
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int r = e * 8 + f;
      double* p = T ? out + ((i / T) * R + r) * (int64_t)T + i % T : out + (int64_t)r * n + i;
      __builtin_nontemporal_store(v[f] + e, p);
    }
    __builtin_nontemporal_store((uint8_t)1, aout + (int64_t)e * n + i);
  }
}

template <class F>
static float timeit(F launch, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

template <int E>
static int run(int64_t n, int nbuf) {
  constexpr int T = 32768;
  n = n / T * T;                                  // whole tiles
  const int64_t nb = n / kB;
  double* in; uint8_t *ain, *aout;
  CK(hipMalloc(&in, 8 * n * 8)); CK(hipMalloc(&ain, n)); CK(hipMalloc(&aout, (size_t)E * n));
  CK(hipMemset(in, 0, 8 * n * 8)); CK(hipMemset(ain, 1, n));
  double* out[32];
  for (int b = 0; b < nbuf; ++b) CK(hipMalloc(&out[b], (size_t)E * 8 * n * 8));
  const double bytes = (57.0 + 65.0 * E) * n;
  printf("E = %d: %d streams written per workgroup, %.2f GB per buffer, %lld rays\n", E, 9 * E, E * 64.0 * n * 1e-9, (long long)n);
  for (int pass = 0; pass < 2; ++pass)
    for (int b = 0; b < nbuf; ++b) {
      const float rows = timeit([&] { k_pattern<E, 0><<<nb, kB>>>(in, ain, out[b], aout, n); }, 30);
      const float blk = timeit([&] { k_pattern<E, T><<<nb, kB>>>(in, ain, out[b], aout, n); }, 30);
      if (pass) printf("  buffer %2d at %p   rows %.4f ms %6.3f TB/s   blocked %.4f ms %6.3f TB/s\n", b, (void*)out[b], rows,
                       bytes / rows * 1e-9, blk, bytes / blk * 1e-9);
      fflush(stdout);
    }
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  for (int b = 0; b < nbuf; ++b) (void)hipFree(out[b]);
  (void)hipFree(in); (void)hipFree(ain); (void)hipFree(aout);
  return 0;
}

// `tlb_probe map [chunk_GiB] [count]`: the rate of the pattern (E = 4: 32 rows + 4 byte rows filling one chunk) in EVERY
// chunk of `count` consecutive allocations -- the map of fast and slow regions a process meets as it allocates
static int map_regions(double gib, int count) {
  constexpr int E = 4;
  const int64_t n = (int64_t)(gib * (1 << 30) / (E * 65)) / 32768 * 32768;
  const int64_t nb = n / kB;
  double* in; uint8_t* ain;
  CK(hipMalloc(&in, 8 * n * 8)); CK(hipMalloc(&ain, n));
  CK(hipMemset(in, 0, 8 * n * 8)); CK(hipMemset(ain, 1, n));
  static double* out[512];
  int got = 0;
  for (; got < count && got < 512; ++got)
    if (hipMalloc(&out[got], (size_t)E * 65 * n) != hipSuccess) { (void)hipGetLastError(); break; }
  printf("map: %d chunks of %.2f GiB (%lld rays x %d rows each), pattern bytes %.2f GB per launch\n", got, E * 65.0 * n / (1 << 30),
         (long long)n, 8 * E, (57.0 + 65.0 * E) * n * 1e-9);
  const double bytes = (57.0 + 65.0 * E) * n;
  for (int pass = 0; pass < 2; ++pass)
    for (int b = 0; b < got; ++b) {
      uint8_t* aout = reinterpret_cast<uint8_t*>(out[b]) + (size_t)E * 64 * n;
      const float ms = timeit([&] { k_pattern<E, 0><<<nb, kB>>>(in, ain, out[b], aout, n); }, 10);
      if (pass) printf("chunk %3d at %p  %.4f ms  %6.3f TB/s\n", b, (void*)out[b], ms, bytes / ms * 1e-9);
    }
  CK(hipDeviceSynchronize());
  for (int b = 0; b < got; ++b) (void)hipFree(out[b]);
  (void)hipFree(in); (void)hipFree(ain);
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'm') return map_regions(argc > 2 ? atof(argv[2]) : 2.0, argc > 3 ? atoi(argv[3]) : 100);
  const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
  const int nbuf = argc > 2 ? atoi(argv[2]) : 12;
  if (n <= 0 || n > 50000000 || nbuf < 1 || nbuf > 32) { printf("arguments out of range\n"); return 2; }
  if (run<4>(n, nbuf)) return 1;
  if (run<8>(n, nbuf)) return 1;
  return 0;
}
