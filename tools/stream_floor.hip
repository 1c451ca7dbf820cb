// Memory floor of the tracing kernel's access pattern, without any arithmetic: one workgroup per 256 rays reads 7 fp64
// streams + 1 byte per ray and writes E x (8 fp64 streams + 1 byte).  Same bytes as the fused chain kernel on relay4
// (E = 4: 57 B read + 260 B written per ray), laid out four ways:
//   soa     the shipped layout, data[8][n] per bundle: 8E + 7 separate fp64 streams, 2 KB contiguous per workgroup each
//   tile    data[n/256][8][256]: every workgroup writes ONE contiguous 16 KB block per bundle
//   fill    write-only, contiguous, the same number of bytes as soa moves in total (what the DRAM takes with no reads)
//   copy    half the bytes read, half written, contiguous
// Build: hipcc -O3 --offload-arch=gfx950 tools/stream_floor.hip -o tools/_build/stream_floor ; run: stream_floor [n] [E]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kB = 256;

template <int E, bool NT>
__global__ __launch_bounds__(kB) void k_soa(const double* __restrict__ in, const uint8_t* __restrict__ ain, double* __restrict__ out,
                                            uint8_t* __restrict__ aout, int64_t n, int64_t pitch) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  double v[8];
#pragma unroll
  for (int f = 0; f < 7; ++f) v[f] = __builtin_nontemporal_load(in + f * pitch + i);
  v[7] = (double)ain[i];
#pragma unroll
  for (int e = 0; e < E; ++e) {
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      double* p = out + ((int64_t)e * 8 + f) * pitch + i;
      if (NT) __builtin_nontemporal_store(v[f] + e, p); else *p = v[f] + e;
    }
    if (NT) __builtin_nontemporal_store((uint8_t)1, aout + (int64_t)e * n + i); else aout[(int64_t)e * n + i] = 1;
  }
}

// the same pattern from workgroups of WG threads (WG rays): fewer, larger workgroups narrow the window of addresses in
// flight per CU at the same number of resident waves
template <int E, int WG>
__global__ __launch_bounds__(WG) void k_soa_wg(const double* __restrict__ in, const uint8_t* __restrict__ ain, double* __restrict__ out,
                                               uint8_t* __restrict__ aout, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x;
  if (i >= n) return;
  double v[8];
#pragma unroll
  for (int f = 0; f < 7; ++f) v[f] = __builtin_nontemporal_load(in + f * n + i);
  v[7] = (double)ain[i];
#pragma unroll
  for (int e = 0; e < E; ++e) {
#pragma unroll
    for (int f = 0; f < 8; ++f) __builtin_nontemporal_store(v[f] + e, out + ((int64_t)e * 8 + f) * n + i);
    __builtin_nontemporal_store((uint8_t)1, aout + (int64_t)e * n + i);
  }
}

template <int E, int TILE>
__global__ __launch_bounds__(kB) void k_tile(const double* __restrict__ in, const uint8_t* __restrict__ ain, double* __restrict__ out,
                                             uint8_t* __restrict__ aout, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  const int64_t t = i / TILE, l = i % TILE;
  double v[8];
#pragma unroll
  for (int f = 0; f < 7; ++f) v[f] = __builtin_nontemporal_load(in + (t * 8 + f) * TILE + l);
  v[7] = (double)ain[i];
#pragma unroll
  for (int e = 0; e < E; ++e) {
#pragma unroll
    for (int f = 0; f < 8; ++f) __builtin_nontemporal_store(v[f] + e, out + (int64_t)e * 8 * n + (t * 8 + f) * TILE + l);
    __builtin_nontemporal_store((uint8_t)1, aout + (int64_t)e * n + i);
  }
}

// the pattern of the FUSED kernel (trace + detector read-out in one launch): + one 8-byte weight stream in, + three 8-byte
// read-out streams out (X, Y, optical path): 65 B read + (65 E + 24) B written per ray
template <int E>
__global__ __launch_bounds__(kB) void k_soa_ro(const double* __restrict__ in, const uint8_t* __restrict__ ain, double* __restrict__ out,
                                               uint8_t* __restrict__ aout, double* __restrict__ ro, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  double v[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) v[f] = __builtin_nontemporal_load(in + f * n + i);     // 7 state streams + the weight
  v[7] += (double)ain[i];
#pragma unroll
  for (int e = 0; e < E; ++e) {
#pragma unroll
    for (int f = 0; f < 8; ++f) __builtin_nontemporal_store(v[f] + e, out + ((int64_t)e * 8 + f) * n + i);
    __builtin_nontemporal_store((uint8_t)1, aout + (int64_t)e * n + i);
  }
#pragma unroll
  for (int f = 0; f < 3; ++f) __builtin_nontemporal_store(v[f] + 9.0, ro + (int64_t)f * n + i);
}

// variants of the 8-byte pattern that take it apart: MODE 0 = loads only (the 7 + 1 input streams, one dummy store per
// workgroup), 1 = plain (cached) loads instead of non-temporal ones, 2 = the stores do not depend on the loads (issued
// first), 3 = the loads come from a 57-KB block that stays in cache (stores as in the pattern)
template <int E, int MODE>
__global__ __launch_bounds__(kB) void k_soa_parts(const double* __restrict__ in, const uint8_t* __restrict__ ain, double* __restrict__ out,
                                                  uint8_t* __restrict__ aout, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  double v[8];
  if (MODE == 2) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma unroll
      for (int f = 0; f < 8; ++f) __builtin_nontemporal_store((double)(e + f), out + ((int64_t)e * 8 + f) * n + i);
      __builtin_nontemporal_store((uint8_t)1, aout + (int64_t)e * n + i);
    }
  }
  const int64_t li = (MODE == 3) ? (i & 1023) : i;
#pragma unroll
  for (int f = 0; f < 7; ++f) v[f] = (MODE == 1 || MODE == 3) ? in[f * n + li] : __builtin_nontemporal_load(in + f * n + li);
  v[7] = (double)ain[li];
  if (MODE == 0 || MODE == 2) {
    double s = 0;
#pragma unroll
    for (int f = 0; f < 8; ++f) s += v[f];
    if (s == -1.2345e300) out[i] = s;      // keeps the loads alive, never true
    return;
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
#pragma unroll
    for (int f = 0; f < 8; ++f) __builtin_nontemporal_store(v[f] + e, out + ((int64_t)e * 8 + f) * n + i);
    __builtin_nontemporal_store((uint8_t)1, aout + (int64_t)e * n + i);
  }
}

// two adjacent rays per lane: 16-byte loads and stores, WG threads per workgroup (2 * WG rays)
template <int E, int WG, bool ALIVE>
__global__ __launch_bounds__(WG) void k_soa16(const double* __restrict__ in, const uint8_t* __restrict__ ain, double* __restrict__ out,
                                              uint8_t* __restrict__ aout, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * WG + threadIdx.x) * 2;
  if (i >= n) return;
  double2 v[8];
#pragma unroll
  for (int f = 0; f < 7; ++f) {
    v[f].x = __builtin_nontemporal_load(in + f * n + i);
    v[f].y = __builtin_nontemporal_load(in + f * n + i + 1);
  }
  v[7] = v[0];
  unsigned short a = 0x0101;
  if (ALIVE) { a = *(const unsigned short*)(ain + i); v[7].x += a; }
#pragma unroll
  for (int e = 0; e < E; ++e) {
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      double* p = out + ((int64_t)e * 8 + f) * n + i;
      __builtin_nontemporal_store(v[f].x + e, p);
      __builtin_nontemporal_store(v[f].y + e, p + 1);
    }
    if (ALIVE) __builtin_nontemporal_store(a, (unsigned short*)(aout + (int64_t)e * n + i));
  }
}

// the 8-byte pattern without the alive bytes
template <int E>
__global__ __launch_bounds__(kB) void k_soa_noalive(const double* __restrict__ in, double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  double v[8];
#pragma unroll
  for (int f = 0; f < 7; ++f) v[f] = __builtin_nontemporal_load(in + f * n + i);
  v[7] = v[0];
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int f = 0; f < 8; ++f) __builtin_nontemporal_store(v[f] + e, out + ((int64_t)e * 8 + f) * n + i);
}

// write-only, 8 bytes per lane, one workgroup per 2 KB
__global__ __launch_bounds__(kB) void k_fill8(double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i < n) __builtin_nontemporal_store(1.0, out + i);
}

// write-only, S streams of 8 bytes per lane from one workgroup (the pattern's stores without its loads)
template <int S>
__global__ __launch_bounds__(kB) void k_fill8_streams(double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int f = 0; f < S; ++f) __builtin_nontemporal_store(1.0 + f, out + (int64_t)f * n + i);
}

__global__ __launch_bounds__(kB) void k_fill(double2* __restrict__ out, int64_t n2) {
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kB)
    __builtin_nontemporal_store(make_double2(1.0, 2.0).x, &out[i].x), __builtin_nontemporal_store(2.0, &out[i].y);
}

// one workgroup per 4 KB chunk, like the streaming kernels (no grid-stride loop)
__global__ __launch_bounds__(kB) void k_fill_wg(double* __restrict__ out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * kB + threadIdx.x) * 2;
  if (i + 1 < n) { __builtin_nontemporal_store(1.0, out + i); __builtin_nontemporal_store(2.0, out + i + 1); }
}

__global__ __launch_bounds__(kB) void k_copy(const double* __restrict__ in, double* __restrict__ out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * kB + threadIdx.x) * 2;
  if (i + 1 < n) {
    const double a = __builtin_nontemporal_load(in + i), b = __builtin_nontemporal_load(in + i + 1);
    __builtin_nontemporal_store(a, out + i); __builtin_nontemporal_store(b, out + i + 1);
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

template <int E>
static int run(int64_t n) {
  const int64_t nb = (n + kB - 1) / kB;
  n = nb * kB;                                   // whole tiles, so that the tiled variants need no tail
  double *in, *out, *ro; uint8_t *ain, *aout;
  CK(hipMalloc(&ro, 3 * n * 8));
  const int64_t kMaxPad = 1 << 22;   // doubles of extra row pitch tried below (<= 32 MB per row)
  CK(hipMalloc(&in, 8 * (n + kMaxPad) * 8)); CK(hipMalloc(&out, (size_t)E * 8 * (n + kMaxPad) * 8)); CK(hipMalloc(&ain, n)); CK(hipMalloc(&aout, (size_t)E * n));
  CK(hipMemset(in, 0, 8 * n * 8)); CK(hipMemset(ain, 1, n));
  const double bytes = (57.0 + 65.0 * E) * n;
  const int reps = 100;
  auto line = [&](const char* name, float ms, double b) { printf("%-34s E=%d  %8.4f ms  %7.3f TB/s  (%.3f GB)\n", name, E, ms, b / ms * 1e-9, b * 1e-9); fflush(stdout); };
  line("soa  nt stores", timeit([&] { k_soa<E, true><<<nb, kB>>>(in, ain, out, aout, n, n); }, reps), bytes);
  line("soa  plain stores", timeit([&] { k_soa<E, false><<<nb, kB>>>(in, ain, out, aout, n, n); }, reps), bytes);
  line("soa  nt + read-out streams (fused)", timeit([&] { k_soa_ro<E><<<nb, kB>>>(in, ain, out, aout, ro, n); }, reps), bytes + 32.0 * n);
  // the same rows further apart: does the relative placement of the 8E + 7 streams in the channel / bank interleave matter?
  const int64_t pads[] = {64, 512, 8192 + 64, 131072 + 512, 1 << 18, 1 << 19, 1 << 20, (1 << 20) + (1 << 19), 1 << 21, (1 << 21) + 512,
                          3 << 20, kMaxPad};
  line("parts: loads only (57 B/ray)", timeit([&] { k_soa_parts<E, 0><<<nb, kB>>>(in, ain, out, aout, n); }, reps), 57.0 * n);
  line("parts: plain (cached) loads", timeit([&] { k_soa_parts<E, 1><<<nb, kB>>>(in, ain, out, aout, n); }, reps), bytes);
  line("parts: stores first, then loads", timeit([&] { k_soa_parts<E, 2><<<nb, kB>>>(in, ain, out, aout, n); }, reps), bytes);
  line("parts: loads from cache, stores", timeit([&] { k_soa_parts<E, 3><<<nb, kB>>>(in, ain, out, aout, n); }, reps), 65.0 * E * n);
  // fewer resident workgroups per CU (dynamic LDS as the limiter; 160 KB per CU): a narrower window of addresses in flight
  const int ldss[] = {20 * 1024, 32 * 1024, 53 * 1024, 80 * 1024};
  for (int lds : ldss) {
    char name[64];
    snprintf(name, sizeof name, "soa  nt, %d workgroups per CU", 160 * 1024 / lds);
    line(name, timeit([&] { k_soa<E, true><<<nb, kB, lds>>>(in, ain, out, aout, n, n); }, reps), bytes);
  }
  // 1024-thread workgroups: 16 waves per workgroup; 2 / 1 of them per CU through the LDS limiter
  line("soa  nt, 1024-thread WGs", timeit([&] { k_soa_wg<E, 1024><<<(n + 1023) / 1024, 1024>>>(in, ain, out, aout, n); }, reps), bytes);
  line("soa  nt, 1024-thread WGs, 1 per CU", timeit([&] { k_soa_wg<E, 1024><<<(n + 1023) / 1024, 1024, 100 * 1024>>>(in, ain, out, aout, n); }, reps), bytes);
  line("soa  nt, 512-thread WGs, 2 per CU", timeit([&] { k_soa_wg<E, 512><<<(n + 511) / 512, 512, 70 * 1024>>>(in, ain, out, aout, n); }, reps), bytes);
  line("soa  nt, 512-thread WGs, 1 per CU", timeit([&] { k_soa_wg<E, 512><<<(n + 511) / 512, 512, 100 * 1024>>>(in, ain, out, aout, n); }, reps), bytes);
  for (int64_t pad : pads) {
    char name[64];
    snprintf(name, sizeof name, "soa  nt, row pitch + %lld B", (long long)pad * 8);
    line(name, timeit([&] { k_soa<E, true><<<nb, kB>>>(in, ain, out, aout, n, n + pad); }, reps), bytes);
  }
  line("tile 256 rays (16 KB blocks)", timeit([&] { k_tile<E, 256><<<nb, kB>>>(in, ain, out, aout, n); }, reps), bytes);
  line("tile 64 rays (4 KB blocks)", timeit([&] { k_tile<E, 64><<<nb, kB>>>(in, ain, out, aout, n); }, reps), bytes);
  line("soa  16 B per lane, 512 rays/WG", timeit([&] { k_soa16<E, 256, true><<<(n / 2 + 255) / 256, 256>>>(in, ain, out, aout, n); }, reps), bytes);
  line("soa  16 B per lane, 256 rays/WG", timeit([&] { k_soa16<E, 128, true><<<(n / 2 + 127) / 128, 128>>>(in, ain, out, aout, n); }, reps), bytes);
  line("soa  16 B per lane, no alive bytes", timeit([&] { k_soa16<E, 128, false><<<(n / 2 + 127) / 128, 128>>>(in, ain, out, aout, n); }, reps), (56.0 + 64.0 * E) * n);
  line("soa   8 B per lane, no alive bytes", timeit([&] { k_soa_noalive<E><<<nb, kB>>>(in, out, n); }, reps), (56.0 + 64.0 * E) * n);
  line("fill  8 B per lane, 8E streams", timeit([&] { k_fill8_streams<8 * E><<<nb, kB>>>(out, n); }, reps), 64.0 * E * n);
  line("fill  8 B per lane, 1 stream", timeit([&] { k_fill8<<<nb * 8 * E, kB>>>(out, n * 8 * E); }, reps), 64.0 * E * n);
  const int64_t nd = (int64_t)(bytes / 8) & ~1LL;
  hipFree(out);
  CK(hipMalloc(&out, (size_t)nd * 8));
  {
    line("fill grid-stride 2048 WGs", timeit([&] { k_fill<<<2048, kB>>>((double2*)out, nd / 2); }, reps), nd * 8.0);
    line("fill one WG per 4 KB", timeit([&] { k_fill_wg<<<(nd / 2 + kB - 1) / kB, kB>>>(out, nd); }, reps), nd * 8.0);
    const int64_t nc = nd / 2 & ~1LL;
    line("copy one WG per 4 KB", timeit([&] { k_copy<<<(nc / 2 + kB - 1) / kB, kB>>>(out, out + nc, nc); }, reps), nc * 16.0);
  }
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  hipFree(in); hipFree(out); hipFree(ain); hipFree(aout); hipFree(ro);
  return 0;
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
  if (n <= 0 || n > 200000000) { printf("n out of range\n"); return 2; }
  for (int pass = 0; pass < 2; ++pass) {       // the first pass also brings the clocks up; read the second
    printf("pass %d\n", pass);
    if (run<1>(n)) return 1;
    if (run<4>(n)) return 1;
    if (n <= 50000000 && run<8>(n)) return 1;
  }
  return 0;
}
