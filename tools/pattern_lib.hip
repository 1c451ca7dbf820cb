// The bare access pattern of a fused chain launch as a C-callable library (for tools/pitch_probe.py --floor): 7 fp64 rows + 1
// byte row read, E x (8 fp64 rows + 1 byte row) written per ray, no arithmetic, into the SAME views a real launch writes.
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/pattern_lib.hip -o tools/_build/libpattern.so
#include <hip/hip_runtime.h>
#include <cstdint>

struct View { double* row[8]; uint8_t* alive; };       // the layout of ArtBundleView (include/art_hip.h)
struct Views { View v[8]; };

template <int E>
__global__ __launch_bounds__(256) void k_bare(const View in, const Views out, const int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v[8];
#pragma unroll
  for (int f = 0; f < 7; ++f) v[f] = __builtin_nontemporal_load(in.row[f] + i);
  v[7] = (double)in.alive[i];
#pragma unroll
  for (int e = 0; e < E; ++e) {
#pragma unroll
    for (int f = 0; f < 8; ++f) __builtin_nontemporal_store(v[f] + e, out.v[e].row[f] + i);
    __builtin_nontemporal_store((uint8_t)1, out.v[e].alive + i);
  }
}

extern "C" int pattern_launch(const View* in, const View* outs, int n_elems, int64_t n, void* stream) {
  Views o{};
  for (int e = 0; e < n_elems && e < 8; ++e) o.v[e] = outs[e];
  const dim3 g((unsigned)((n + 255) / 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
  switch (n_elems) {
    case 1: hipLaunchKernelGGL(k_bare<1>, g, b, 0, s, *in, o, n); break;
    case 2: hipLaunchKernelGGL(k_bare<2>, g, b, 0, s, *in, o, n); break;
    case 3: hipLaunchKernelGGL(k_bare<3>, g, b, 0, s, *in, o, n); break;
    case 4: hipLaunchKernelGGL(k_bare<4>, g, b, 0, s, *in, o, n); break;
    case 8: hipLaunchKernelGGL(k_bare<8>, g, b, 0, s, *in, o, n); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
