// art_kernels.hip -- gfx950 kernels + the C ABI of libart_hip.so (include/art_hip.h).
//
// One ray per lane, SoA fp64 streams read and written with coalesced, non-temporal 8-byte accesses (64 lanes x
// 8 B = 512 B per wave instruction and per array; 16 B per lane in the read-out), element descriptors in kernel
// arguments or in a device-resident scene table (either way scalar loads -> SGPRs, broadcast for free), Zernike
// coefficient tables read through the scalar cache as well (wave-uniform; -DART_ZERN_LDS stages them in LDS instead),
// wavefront ballot on the torus Newton loop.  No MFMA: the path is streaming fp64 VALU work against HBM.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "art_device.h"
#include "art_scene.h"

// The -DART_ZERN_LDS comparison build keeps a persistent grid whose last pass may leave part of a workgroup idle: no
// workgroup barriers there, i.e. the 8-byte store path.
#if defined(ART_ZERN_LDS) && !defined(ART_STORE_DIRECT)
#define ART_STORE_DIRECT 1
#endif

namespace {

constexpr int kBlock = 256;          // 4 waves per workgroup
constexpr int kMaxBlocks = 256 * 8;  // reductions: 256 CUs x 8 workgroups, grid-stride beyond that (one partial each)
constexpr int kReadoutBlocks = kMaxBlocks;  // fused read-out: persistent workgroups -- every workgroup ends with a 24-slot
                                           // reduction, so MORE workgroups cost more (16384: 244 us instead of 176 us)

thread_local char g_err[512] = "";

int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
int fail_hip(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return ART_ERR_HIP;
}

// The two trace kernels get one workgroup per 256 rays, NOT a persistent grid with a grid-stride loop: measured on
// relay4, 1e7 rays, the fused kernel takes 0.686 ms with 2048 workgroups and 0.653 ms with 39063 (per-element kernel
// 0.243 -> 0.222 ms; at 1e8 rays 7.33 -> 6.35 ms).  Consecutive workgroups then sweep every stream linearly (DRAM
// pages, TLB), and the hardware's workgroup dispatch hides latency at least as well as a software prefetch did.
// It is not a general rule (all per 1e7 rays, launches back to back): the bundle transform also gains (263 -> 226 us),
// the source generator (plain stores behind heavy trigonometry) loses (138 -> 158 us) and the fused read-out loses
// (every workgroup ends in a 24-slot reduction), so those keep the persistent grid_for().
inline int grid_stream(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > (int64_t)1 << 22) b = (int64_t)1 << 22;   // 2^28 rays per launch at most: never reached
  return (int)b;
}

// Workgroup -> tile mapping of the streaming kernels (`xmap`, a kernel argument).  The hardware deals consecutive
// workgroup ids round-robin to the 8 XCDs, so with the identity mapping each XCD's L2 sees every 8th 2-KiB piece of
// every stream.  xmap = k > 0 hands each XCD runs of 2^(k-1) CONSECUTIVE tiles instead (groups of 8 * 2^(k-1) tiles are
// shared out among the XCDs); xmap < 0 gives every XCD one contiguous eighth of the launch; 0 = identity.  The grid is
// rounded up to a whole number of groups (grid_stream_mapped); tiles beyond the bundle fall outside every buffer
// descriptor and do nothing.  Measured in DESIGN.md 5 (round 2).
__device__ __forceinline__ int64_t tile_of(const unsigned b, const unsigned nb, const int xmap) {
  if (xmap == 0) return b;
  if (xmap < 0) return (int64_t)(b & 7u) * (nb >> 3) + (b >> 3);
  const unsigned sh = (unsigned)(xmap - 1), span_sh = sh + 3u;
  const unsigned r = b & ((1u << span_sh) - 1u);
  return (int64_t)(((b >> span_sh) << span_sh) + ((r & 7u) << sh) + (r >> 3));
}

// ART_XCD_MAP: "0" identity (default: no mapping measured faster than run-to-run noise), "-1" eighths, "k" runs of
// 2^(k-1) tiles per XCD
inline int xcd_map() {
  static const int v = [] {
    const char* e = getenv("ART_XCD_MAP");
    const int x = e ? atoi(e) : 0;
    return (x < -1 || x > 8) ? 0 : x;     // runs of at most 128 tiles: the grid is rounded up by < 1024 workgroups
  }();
  return v;
}

inline int grid_stream_mapped(int64_t n, int xmap) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  const int64_t span = (xmap == 0) ? 1 : (xmap < 0 ? 8 : ((int64_t)8 << (xmap - 1)));
  b = (b + span - 1) / span * span;
  return (int)b;   // <= 2^20 + span workgroups for 2^28 rays
}

inline int grid_for(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

// ---- buffer-resource access -----------------------------------------------------------------------------
// Every stream is addressed through a 128-bit buffer descriptor (base, byte count) with the per-lane byte offset
// in a 32-bit VGPR.  The hardware range check gives branch-free predication: a lane whose offset is out of range
// loads 0 / stores nothing.  That is what lets dead rays skip their 8 output stores WITHOUT a branch around the
// stores -- a branch would make the compiler drain every outstanding store (s_waitcnt vmcnt(0)) at the join, between
// the elements of a chain -- and the tail of a launch (slots >= n) needs no branch either.  One launch covers at most 2^28 rays (32-bit byte offsets); the C ABI splits
// larger bundles into several launches.
typedef int v2i32 __attribute__((ext_vector_type(2)));
// Cache-policy bits of the buffer instructions (gfx942/gfx950 aux operand: 1 = sc0, 2 = nt, 16 = sc1).  Every ray
// stream is touched exactly once per kernel, so both directions are marked non-temporal: measured on relay4
// (bench.py, 1e7 rays) 0.937 -> 0.874 ms per step against the default policy; sc0/sc1 combinations made no further
// difference.  The macros exist so that the experiment can be repeated (-DART_LD_AUX=0 -DART_ST_AUX=0).
#ifndef ART_LD_AUX
#define ART_LD_AUX 2
#endif
#ifndef ART_ST_AUX
#define ART_ST_AUX 2
#endif
constexpr int64_t kMaxRaysPerLaunchHw = (int64_t)1 << 28;  // 2^28 rays * 8 B = 2 GiB per stream
// Offset of a store that must be dropped by the range check: 2^31.  num_records never exceeds 2^31 (2^28 rays x 8 B), so
// it is out of range for every descriptor; it is 16-byte aligned, and the component offsets of a 16-byte store (+4, +8,
// +12) cannot wrap around 2^32 into the first slots of the stream (0xFFFFFFFF, used before, relied on that not happening).
constexpr unsigned kDropOffset = 0x80000000u;

struct BundleRsrc {
  __amdgpu_buffer_rsrc_t ox, oy, oz, dx, dy, dz, path, inc, alive;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000);
}
// descriptors of the n slots [first, first + n) of a view
__device__ __forceinline__ BundleRsrc make_rsrc(const ArtBundleView& v, int64_t n, int64_t first = 0) {
  const unsigned b8 = (unsigned)(n * 8), b1 = (unsigned)n;
  BundleRsrc r;
  r.ox = rsrc_of(v.ox + first, b8); r.oy = rsrc_of(v.oy + first, b8); r.oz = rsrc_of(v.oz + first, b8);
  r.dx = rsrc_of(v.dx + first, b8); r.dy = rsrc_of(v.dy + first, b8); r.dz = rsrc_of(v.dz + first, b8);
  r.path = rsrc_of(v.path + first, b8); r.inc = rsrc_of(v.incidence + first, b8);
  r.alive = rsrc_of(v.alive + first, b1);
  return r;
}
__device__ __forceinline__ double ld_f64(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  const v2i32 d = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, ART_LD_AUX);
  double v;
  __builtin_memcpy(&v, &d, 8);
  return v;
}
__device__ __forceinline__ void st_f64(__amdgpu_buffer_rsrc_t rs, unsigned off, double v) {
  v2i32 d;
  __builtin_memcpy(&d, &v, 8);
  __builtin_amdgcn_raw_buffer_store_b64(d, rs, (int)off, 0, ART_ST_AUX);
}
typedef int v4i32 __attribute__((ext_vector_type(4)));
struct D2 { double a, b; };
__device__ __forceinline__ D2 ld_2f64(__amdgpu_buffer_rsrc_t rs, unsigned off) {   // 16 B per lane: two consecutive slots
  const v4i32 d = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, ART_LD_AUX);
  D2 v;
  __builtin_memcpy(&v, &d, 16);
  return v;
}
__device__ __forceinline__ void st_2f64(__amdgpu_buffer_rsrc_t rs, unsigned off, double a, double b) {
  D2 v = {a, b};
  v4i32 d;
  __builtin_memcpy(&d, &v, 16);
  __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)off, 0, ART_ST_AUX);
}

// slot index -> byte offsets; out-of-range slots (i >= n) fall outside every descriptor automatically
__device__ __forceinline__ void load_slot(const BundleRsrc& b, int64_t i, art::Ray& r, uint8_t& alive) {
  const unsigned o1 = (unsigned)i, o8 = o1 * 8u;
  alive = __builtin_amdgcn_raw_buffer_load_b8(b.alive, (int)o1, 0, ART_LD_AUX);
  r.ox = ld_f64(b.ox, o8); r.oy = ld_f64(b.oy, o8); r.oz = ld_f64(b.oz, o8);
  r.dx = ld_f64(b.dx, o8); r.dy = ld_f64(b.dy, o8); r.dz = ld_f64(b.dz, o8);
  r.path = ld_f64(b.path, o8);
}
// The same loads with the DEFAULT cache policy: for an input that other workgroups are about to read again (the shared
// input of a chain-interleaved scene launch) -- a non-temporal load does not leave its line in the caches.
__device__ __forceinline__ double ld_f64_keep(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  const v2i32 d = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
  double v;
  __builtin_memcpy(&v, &d, 8);
  return v;
}
__device__ __forceinline__ D2 ld_2f64_keep(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  const v4i32 d = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
  D2 v;
  __builtin_memcpy(&v, &d, 16);
  return v;
}
__device__ __forceinline__ void load_slot_keep(const BundleRsrc& b, int64_t i, art::Ray& r, uint8_t& alive) {
  const unsigned o1 = (unsigned)i, o8 = o1 * 8u;
  alive = __builtin_amdgcn_raw_buffer_load_b8(b.alive, (int)o1, 0, 0);
  r.ox = ld_f64_keep(b.ox, o8); r.oy = ld_f64_keep(b.oy, o8); r.oz = ld_f64_keep(b.oz, o8);
  r.dx = ld_f64_keep(b.dx, o8); r.dy = ld_f64_keep(b.dy, o8); r.dz = ld_f64_keep(b.dz, o8);
  r.path = ld_f64_keep(b.path, o8);
}
__device__ __forceinline__ void store_slot(const BundleRsrc& b, int64_t i, const art::Ray& r, bool ok) {
  const unsigned o1 = (unsigned)i;
#ifdef ART_DIAG_NOSTORE   // timing-only build: keep the compute, drop the 8 data stores (results are wrong)
  const unsigned o8 = (ok && r.path == -1.2345e300) ? o1 * 8u : kDropOffset;
#else
  const unsigned o8 = ok ? o1 * 8u : kDropOffset;  // dead rays: the range check drops the 8 stores
#endif
  st_f64(b.ox, o8, r.ox); st_f64(b.oy, o8, r.oy); st_f64(b.oz, o8, r.oz);
  st_f64(b.dx, o8, r.dx); st_f64(b.dy, o8, r.dy); st_f64(b.dz, o8, r.dz);
  st_f64(b.path, o8, r.path);
  st_f64(b.inc, o8, r.inc);
#ifdef ART_DIAG_NOALIVE   // timing-only build: slots >= 4096 get no alive store (results of large bundles are wrong)
  __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(ok ? 1 : 0), b.alive, o1 < 4096u ? (int)o1 : (int)kDropOffset, 0, ART_ST_AUX);
#else
  __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(ok ? 1 : 0), b.alive, (int)o1, 0, ART_ST_AUX);
#endif
}

// plain-pointer access for the small kernels (detector, sources)
__device__ __forceinline__ void load_ray(const ArtBundleView& v, int64_t i, art::Ray& r) {
  r.ox = v.ox[i]; r.oy = v.oy[i]; r.oz = v.oz[i];
  r.dx = v.dx[i]; r.dy = v.dy[i]; r.dz = v.dz[i];
  r.path = v.path[i];
}
__device__ __forceinline__ void store_ray(const ArtBundleView& v, int64_t i, const art::Ray& r) {
  v.ox[i] = r.ox; v.oy[i] = r.oy; v.oz[i] = r.oz;
  v.dx[i] = r.dx; v.dy[i] = r.dy; v.dz[i] = r.dz;
  v.path[i] = r.path;
  v.incidence[i] = r.inc;
}

// ---- masked streaming reads of the reduction kernels -------------------------------------------------------------------
// A reduction over the alive rays used to read `alive[i]`, branch, and only then load the slot's data: two dependent memory
// latencies per iteration and ONE iteration's loads in flight per lane (0.42-0.63 of the rate the trace kernels reach,
// profiles/r04_relay4.md).  Now every stream of kU grid-stride iterations is requested UNCONDITIONALLY and up front
// (non-temporal: each byte is used once), and `alive` is applied by SELECTS when the values are folded: 4 x 9 loads in
// flight per lane, no divergent skip.  A lane beyond the end re-reads slot n - 1 (a valid address, a cache hit) and folds
// nothing.  The values of a dead slot are unspecified by contract (possibly NaN): they are dropped by the select, never
// multiplied by 0.  The fold ORDER is unchanged: lane t of workgroup b still folds slots b*256 + t + k*stride for
// k = 0, 1, ... one after the other, so the results are the bits they were.
// MASKED bundles (a third of C3's slots are dead, in one contiguous range): these loops read the dead slots' streams too.
// The two kernels of the loop-list analysis, where that matters (k_analysis_sums, k_analysis_moments), request a slot's
// data at an out-of-range buffer offset when it is dead -- no traffic -- with the alive byte fetched one slot ahead.
// (the pointer is stated to be GLOBAL memory: one that was itself fetched from a table in memory -- a job's bundle view --
// would otherwise be a generic pointer and compile to flat_load, which also occupies the LDS counter)
template <typename T>
__device__ __forceinline__ T ld_nt(const T* p) {
  typedef const T __attribute__((address_space(1)))* gptr_t;
  return __builtin_nontemporal_load((gptr_t)p);
}
template <typename T>
__device__ __forceinline__ void st_nt(T* p, const T v) {
  typedef T __attribute__((address_space(1)))* gptr_t;
  __builtin_nontemporal_store(v, (gptr_t)p);
}
constexpr int kRedUnroll = 4;
// ------------------------------------------------------------------------------------------- reductions
// Deterministic: fixed grid, each lane accumulates its grid-stride slice, wave shuffle tree, LDS across the
// 4 waves, one partial per workgroup into `scratch`, then fold_slot(): one workgroup per statistic folds the
// partials in a fixed order.
constexpr int kRedBlocks = 1024;
constexpr int kRedSlots = 16;
constexpr int kReadoutSlots = 24;   // art_detector_readout / fused chain read-out statistics
constexpr int kUsedSlots = 22;      // slots 22, 23 are reserved (always 0): the fused tail writes no partials for them

enum RedOp { RSUM = 0, RMIN = 1, RMAX = 2 };

__device__ __forceinline__ double wave_reduce(double v, int op) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(v, off, 64);
    v = (op == RSUM) ? v + o : (op == RMIN ? fmin(v, o) : fmax(v, o));
  }
  return v;
}

// ---- 24 statistics x 64 lanes -> 24 wave totals, for the fused read-out tail (once per wave of 64 rays, so it has to
// be cheap: 24 shuffle trees through ds_bpermute cost more than tracing the rays, 24 DPP prefix trees still +50 %).
// Transpose through a wave-private LDS tile instead: in three passes of 8 statistics every lane parks its 8 values
// ([stat][lane], conflict-free), then lane l folds 8 of the 64 values of statistic l/8 (stride-8 reads) and the 8
// lanes of a statistic finish with a 3-step DPP butterfly (quad_perm xor 1, xor 2, row_half_mirror).  ~100
// instructions per lane instead of ~900; no barrier (same-wave LDS operations are ordered), no VMEM.
template <int CTRL>
__device__ __forceinline__ double dpp_xchg(const double v) {
  int2 s, r;
  __builtin_memcpy(&s, &v, 8);
  // every lane has a source under these controls: no "old" value to preserve (bound_ctrl), so no copy before the move
  r.x = __builtin_amdgcn_update_dpp(0, s.x, CTRL, 0xf, 0xf, true);
  r.y = __builtin_amdgcn_update_dpp(0, s.y, CTRL, 0xf, 0xf, true);
  double d;
  __builtin_memcpy(&d, &r, 8);
  return d;
}
// fmin() on values that come out of LDS compiles to v_max(x, x) (quieting a signalling NaN) + v_min: the bare
// instruction does the same fold in one
__device__ __forceinline__ double min_raw(const double a, const double b) {
  double d;
  asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
constexpr int kTileStride = 72;   // doubles per statistic row of the tile (64 + padding against bank conflicts)
// Which statistic sits where: passes 0 and 1 carry the 16 sums, pass 2 the minima and -- negated, max(x) = -min(-x) --
// the maxima, so that every pass folds with ONE compile-time operator (a per-lane operator compiles to branches).
// Entries >= 24 are padding.
constexpr int kPassSlot[3][8] = {{0, 1, 6, 7, 8, 9, 10, 11}, {16, 17, 18, 19, 20, 21, 14, 15},
                                 {2, 4, 12, 3, 5, 13, 24, 24}};
// The per-wave partials go to the scratch area in THIS order (row = pass * 8 + j, rows 22 and 23 unused), so that the
// tail's three stores need no slot look-up; the fold kernels translate (row_of_slot).
constexpr int row_of_slot(const int slot) {
  for (int p = 0; p < 3; ++p)
    for (int j = 0; j < 8; ++j)
      if (kPassSlot[p][j] == slot) return p * 8 + j;
  return -1;
}

// tile: 8 * kTileStride doubles owned by this wave.  After the call, lanes with (lane & 7) == 0 hold in out[pass] the
// wave total of statistic kPassSlot[pass][lane >> 3] (= row pass * 8 + (lane >> 3) of the scratch area).
__device__ __forceinline__ void wave_reduce24(const double (&acc)[kReadoutSlots], double* tile, const int l,
                                              double (&out)[3]) {
  const int stat = l >> 3, part = l & 7;   // l = lane of the wave
#pragma unroll
  for (int pass = 0; pass < 3; ++pass) {
    // The lanes exchange data through the tile: wave-level barriers (no instruction, a scheduling fence for the compiler)
    // keep the stores of a pass in front of its row loads and the row loads in front of the next pass's stores.  The
    // hardware executes a wave's LDS operations in order; this pins the compiler to the same order.
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = kPassSlot[pass][j];
      tile[j * kTileStride + l] = (g >= kReadoutSlots) ? INFINITY : ((pass == 2 && j >= 3) ? -acc[g] : acc[g]);
    }
    __builtin_amdgcn_wave_barrier();
    const double* row = tile + stat * kTileStride + part;
    double v = row[0];
    if (pass < 2) {
#pragma unroll
      for (int k = 1; k < 8; ++k) v += row[8 * k];
      v += dpp_xchg<0xB1>(v);    // quad_perm:[1,0,3,2]  lane ^ 1
      v += dpp_xchg<0x4E>(v);    // quad_perm:[2,3,0,1]  lane ^ 2
      v += dpp_xchg<0x141>(v);   // row_half_mirror      lane i <-> 7 - i of its group of 8
    } else {
#pragma unroll
      for (int k = 1; k < 8; ++k) v = min_raw(v, row[8 * k]);
      v = min_raw(v, dpp_xchg<0xB1>(v));
      v = min_raw(v, dpp_xchg<0x4E>(v));
      v = min_raw(v, dpp_xchg<0x141>(v));
      v = (stat >= 3) ? -v : v;  // the maxima were folded as minima of their negatives
    }
    out[pass] = v;
  }
}

// The LITE tail's reduction: the two sums (rows 0, 1 of pass 0) and pass 2 (minima / maxima) of wave_reduce24; out[1] = 0
// and the lanes of pass 0's other rows hold 0, so that the partial statistics written behind it have the full layout.
__device__ __forceinline__ void wave_reduce_lite(const double (&acc)[kReadoutSlots], double* tile, const int l,
                                                 double (&out)[3]) {
  const int stat = l >> 3, part = l & 7;
  __builtin_amdgcn_wave_barrier();
  tile[0 * kTileStride + l] = acc[0];
  tile[1 * kTileStride + l] = acc[1];
  __builtin_amdgcn_wave_barrier();
  {
    const double* row = tile + (stat & 1) * kTileStride + part;
    double v = row[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) v += row[8 * k];
    v += dpp_xchg<0xB1>(v);
    v += dpp_xchg<0x4E>(v);
    v += dpp_xchg<0x141>(v);
    out[0] = (stat < 2) ? v : 0.0;
  }
  out[1] = 0.0;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int g = kPassSlot[2][j];
    tile[j * kTileStride + l] = (g >= kReadoutSlots) ? INFINITY : ((j >= 3) ? -acc[g] : acc[g]);
  }
  __builtin_amdgcn_wave_barrier();
  {
    const double* row = tile + stat * kTileStride + part;
    double v = row[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) v = min_raw(v, row[8 * k]);
    v = min_raw(v, dpp_xchg<0xB1>(v));
    v = min_raw(v, dpp_xchg<0x4E>(v));
    v = min_raw(v, dpp_xchg<0x141>(v));
    out[2] = (stat >= 3) ? -v : v;
  }
}

// ---- the nine sums of the analysis' pass (1): count, sum point (3), sum vector (3), sum w, sum path ---------------------
// ONE canonical fold order, whoever forms them -- the tail of a tracing launch (the ray is still in registers: ArtChainReadout.sums),
// either kernel body, or k_analysis_sums re-reading a bundle: per TILE of 256 consecutive slots, four runs of 64 slots
// are folded by the LDS transpose below (lane l parks its 8 values, folds 8 of the 64 values of sum l/8, then a 3-step
// DPP butterfly), the four run totals are added in slot order, and the per-tile partials (row-major: scratch[row * ntiles
// + tile]) are folded by fold_range.  The count is a popcount of ballots (exact in any order).  So a bundle's sums -- and
// everything the analysis derives from them -- are the same bits whether they rode on the trace or were taken afterwards.
constexpr int kSumRows = 9;
// every lane has parked its values in `tile` rows 0..7 (row j = sum j + 1, column = slot of the run); returns, in all 8 lanes
// of group l >> 3, the run's total of sum (l >> 3) + 1
__device__ __forceinline__ double run_fold8(const double* tile, const int l) {
  const double* row = tile + (l >> 3) * kTileStride + (l & 7);
  double v = row[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) v += row[8 * k];
  v += dpp_xchg<0xB1>(v);
  v += dpp_xchg<0x4E>(v);
  v += dpp_xchg<0x141>(v);
  return v;
}
// one ray per lane, a wave = one run: -> the run's total of sum (l >> 3) + 1
__device__ __forceinline__ double run_sums8(const double (&v)[8], double* tile, const int l) {
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 8; ++j) tile[j * kTileStride + l] = v[j];
  __builtin_amdgcn_wave_barrier();
  return run_fold8(tile, l);
}
// the masked values of one ray
__device__ __forceinline__ void sums_values(double (&v)[8], const bool live, const art::Ray& r, const double w) {
  v[0] = live ? r.ox : 0.0; v[1] = live ? r.oy : 0.0; v[2] = live ? r.oz : 0.0;
  v[3] = live ? r.dx : 0.0; v[4] = live ? r.dy : 0.0; v[5] = live ? r.dz : 0.0;
  v[6] = live ? w : 0.0; v[7] = live ? r.path : 0.0;
}
// a workgroup of 4 waves = one tile (one ray per lane): run totals -> the tile's partial, row-major in `rows`.
// s_run: 4 x kSumRows doubles of LDS.  Bare s_barrier: __syncthreads() would also wait for the stores in flight.
// Tiles beyond `ntiles` (a grid rounded up by the tile mapping) store nothing.
__device__ __forceinline__ void tile_sums_store(const double tot, const unsigned long long live_mask, double* s_run,
                                                const unsigned t, double* rows, const int64_t ntiles, const int64_t tile) {
  const int l = t & 63, w = t >> 6;
  if ((l & 7) == 0) s_run[w * kSumRows + 1 + (l >> 3)] = tot;
  if (l == 0) s_run[w * kSumRows] = (double)__popcll(live_mask);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (t < kSumRows && tile < ntiles)
    rows[(int64_t)t * ntiles + tile] = ((s_run[t] + s_run[kSumRows + t]) + s_run[2 * kSumRows + t]) + s_run[3 * kSumRows + t];
}
// Two NEIGHBOURING slots per lane (chain_body2): a wave's 128 slots are two runs -- lanes 0..31 hold run A, lanes 32..63
// run B -- and a workgroup's 512 slots two tiles (waves 0, 1 and waves 2, 3).  Same order as above: the values are parked
// in slot order, one run at a time.
__device__ __forceinline__ void run_sums8_pairs(const double (&v0)[8], const double (&v1)[8], double* tile, const int l,
                                                double& totA, double& totB) {
  const int col = 2 * (l & 31);
  __builtin_amdgcn_wave_barrier();
  if (l < 32) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { tile[j * kTileStride + col] = v0[j]; tile[j * kTileStride + col + 1] = v1[j]; }
  }
  __builtin_amdgcn_wave_barrier();
  totA = run_fold8(tile, l);
  __builtin_amdgcn_wave_barrier();
  if (l >= 32) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { tile[j * kTileStride + col] = v0[j]; tile[j * kTileStride + col + 1] = v1[j]; }
  }
  __builtin_amdgcn_wave_barrier();
  totB = run_fold8(tile, l);
}
// s_run: 8 x kSumRows doubles (the workgroup's eight runs); tiles 2 * pair_tile and 2 * pair_tile + 1
__device__ __forceinline__ void tile_sums_store_pairs(const double totA, const double totB, const unsigned long long m0,
                                                      const unsigned long long m1, double* s_run, const unsigned t,
                                                      double* rows, const int64_t ntiles, const int64_t pair_tile) {
  const int l = t & 63, w = t >> 6;
  if ((l & 7) == 0) {
    s_run[(2 * w) * kSumRows + 1 + (l >> 3)] = totA;
    s_run[(2 * w + 1) * kSumRows + 1 + (l >> 3)] = totB;
  }
  if (l == 0) {
    s_run[(2 * w) * kSumRows] = (double)(__popcll(m0 & 0xffffffffull) + __popcll(m1 & 0xffffffffull));
    s_run[(2 * w + 1) * kSumRows] = (double)(__popcll(m0 >> 32) + __popcll(m1 >> 32));
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  const int h = (int)(t >> 6);            // wave 0 stores the first tile, wave 1 the second
  const int row = (int)(t & 63);
  const int64_t tile = 2 * pair_tile + h;
  if (h < 2 && row < kSumRows && tile < ntiles) {
    const double* q = s_run + 4 * h * kSumRows + row;
    rows[(int64_t)row * ntiles + tile] = ((q[0] + q[kSumRows]) + q[2 * kSumRows]) + q[3 * kSumRows];
  }
}

template <int NS>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NS], const int (&ops)[NS], double* dst) {
  __shared__ double s[kBlock / 64][NS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const double v = wave_reduce(acc[k], ops[k]);
    if (lane == 0) s[w][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    const int k = threadIdx.x;
    double v = s[0][k];
    for (int j = 1; j < kBlock / 64; ++j)
      v = (ops[k] == RSUM) ? v + s[j][k] : (ops[k] == RMIN ? fmin(v, s[j][k]) : fmax(v, s[j][k]));
    dst[k] = v;
  }
}

// The same with the operator of slot k given by a function (computed, not looked up: a run-time indexed local array of
// operators lives in scratch memory).
template <int NS, typename F>
__device__ __forceinline__ void block_reduce_store_f(double (&acc)[NS], F op_of, double* dst) {
  __shared__ double s[kBlock / 64][NS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const double v = wave_reduce(acc[k], op_of(k));
    if (lane == 0) s[w][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    const int k = threadIdx.x;
    const int op = op_of(k);
    double v = s[0][k];
    for (int j = 1; j < kBlock / 64; ++j)
      v = (op == RSUM) ? v + s[j][k] : (op == RMIN ? fmin(v, s[j][k]) : fmax(v, s[j][k]));
    dst[k] = v;
  }
}

// Final fold of per-block partials laid out [block][ns]: launched with ns blocks, block k folds slot k in a fixed
// order (thread t takes partials t, t+256, ...; then the shuffle tree) — deterministic, and ns blocks work in
// parallel instead of one block walking the whole table.
__device__ __forceinline__ void fold_slot(const double* scratch, const int nblocks, const int ns, const int op_k,
                                          double* out) {
  const int k = blockIdx.x;
  const int op[1] = {op_k};
  const double ident = (op_k == RSUM) ? 0.0 : (op_k == RMIN ? INFINITY : -INFINITY);
  double acc[1] = {ident};
  constexpr int kU = 8;     // loads in flight per thread (one at a time they are that many memory latencies in a row)
  for (int blk = threadIdx.x; blk < nblocks; blk += kBlock * kU) {
    double v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) v[u] = (blk + u * kBlock < nblocks) ? scratch[(int64_t)(blk + u * kBlock) * ns + k] : ident;
#pragma unroll
    for (int u = 0; u < kU; ++u)
      acc[0] = (op_k == RSUM) ? acc[0] + v[u] : (op_k == RMIN ? fmin(acc[0], v[u]) : fmax(acc[0], v[u]));
  }
  block_reduce_store<1>(acc, op, out + k);
}

// The 24 read-out statistics (layout: art_hip.h, art_detector_readout) of one ray folded into `acc`; dead rays add the
// reduction identities through selects (no divergent skip).
__device__ __forceinline__ void readout_accumulate(double (&acc)[kReadoutSlots], const bool live, const double x,
                                                   const double y, const double o, const double wi, const bool has_w,
                                                   const double cx, const double cy, const double co) {
  const double ww = live ? (has_w ? wi : 1.0) : 0.0, one = live ? 1.0 : 0.0;
  const double xs = live ? x : 0.0, ys = live ? y : 0.0, os = live ? o : 0.0;
  acc[0] += one; acc[1] += os;
  acc[2] = fmin(acc[2], live ? x : INFINITY); acc[3] = fmax(acc[3], live ? x : -INFINITY);
  acc[4] = fmin(acc[4], live ? y : INFINITY); acc[5] = fmax(acc[5], live ? y : -INFINITY);
  acc[6] += xs; acc[7] += ys;
  acc[8] += ww; acc[9] = fma(ww, xs, acc[9]); acc[10] = fma(ww, ys, acc[10]); acc[11] = fma(ww, os, acc[11]);
  acc[12] = fmin(acc[12], live ? o : INFINITY); acc[13] = fmax(acc[13], live ? o : -INFINITY);
  const double ex = live ? x - cx : 0.0, ey = live ? y - cy : 0.0, eo = live ? o - co : 0.0;
  acc[16] = fma(ex, ex, acc[16]); acc[17] = fma(ey, ey, acc[17]); acc[18] = fma(eo, eo, acc[18]);
  acc[19] = fma(ww * ex, ex, acc[19]); acc[20] = fma(ww * ey, ey, acc[20]); acc[21] = fma(ww * eo, eo, acc[21]);
}
// The same for the fused tail, where a lane holds exactly ONE ray: the statistics are assigned, not added to their
// identities (0.0 + x is not x for the compiler: -0.0), which is 14 instructions less per ray.
__device__ __forceinline__ void readout_single(double (&acc)[kReadoutSlots], const bool live, const double x,
                                               const double y, const double o, const double wi, const bool has_w,
                                               const double cx, const double cy, const double co) {
  const double ww = live ? (has_w ? wi : 1.0) : 0.0;
  const double xs = live ? x : 0.0, ys = live ? y : 0.0, os = live ? o : 0.0;
  acc[0] = live ? 1.0 : 0.0; acc[1] = os;
  acc[2] = live ? x : INFINITY; acc[3] = live ? x : -INFINITY;
  acc[4] = live ? y : INFINITY; acc[5] = live ? y : -INFINITY;
  acc[6] = xs; acc[7] = ys;
  acc[8] = ww; acc[9] = ww * xs; acc[10] = ww * ys; acc[11] = ww * os;
  acc[12] = live ? o : INFINITY; acc[13] = live ? o : -INFINITY;
  acc[14] = 0.0; acc[15] = 0.0;
  const double ex = live ? x - cx : 0.0, ey = live ? y - cy : 0.0, eo = live ? o - co : 0.0;
  acc[16] = ex * ex; acc[17] = ey * ey; acc[18] = eo * eo;
  acc[19] = ww * ex * ex; acc[20] = ww * ey * ey; acc[21] = ww * eo * eo;
  acc[22] = 0.0; acc[23] = 0.0;
}
// LITE tail (ArtChainReadout.lite): count, sum of paths, bounding box and path range only -- 8 of the 22 statistics, no
// weights.  The other slots read 0.
__device__ __forceinline__ void readout_single_lite(double (&acc)[kReadoutSlots], const bool live, const double x,
                                                    const double y, const double o) {
  acc[0] = live ? 1.0 : 0.0; acc[1] = live ? o : 0.0;
  acc[2] = live ? x : INFINITY; acc[3] = live ? x : -INFINITY;
  acc[4] = live ? y : INFINITY; acc[5] = live ? y : -INFINITY;
  acc[12] = live ? o : INFINITY; acc[13] = live ? o : -INFINITY;
}
__device__ __forceinline__ void readout_accumulate_lite(double (&acc)[kReadoutSlots], const bool live, const double x,
                                                        const double y, const double o) {
  acc[0] += live ? 1.0 : 0.0; acc[1] += live ? o : 0.0;
  acc[2] = fmin(acc[2], live ? x : INFINITY); acc[3] = fmax(acc[3], live ? x : -INFINITY);
  acc[4] = fmin(acc[4], live ? y : INFINITY); acc[5] = fmax(acc[5], live ? y : -INFINITY);
  acc[12] = fmin(acc[12], live ? o : INFINITY); acc[13] = fmax(acc[13], live ? o : -INFINITY);
}
#define ART_READOUT_OPS {RSUM, RSUM, RMIN, RMAX, RMIN, RMAX, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, \
                         RMIN, RMAX, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM}

// ------------------------------------------------------------------------------------------- trace kernels
// The descriptor travels in a one-element array and is indexed with blockIdx.y (always 0): a dynamic index makes the
// compiler fetch descriptor fields with scalar loads where they are used instead of keeping all ~60 of them (plus
// 72 SGPRs of buffer descriptors) live across the ray loop, which overflowed the SGPR file into VGPR lanes (200
// v_readlane/v_writelane in the torus kernel).  The fused chain kernel indexes its descriptors by element anyway.
struct ElemArg {
  ArtElementDesc e[1];
};

#ifdef ART_ZERN_LDS
// Comparison build: the dense Zernike tables are staged in LDS once per workgroup and read by every
// lane from the same address, instead of travelling through the scalar cache.  To amortise the staging these kernels
// keep a persistent grid + loop, as in round 1.
constexpr bool kDefectLoop = true;
__device__ __forceinline__ void stage_tables(const double* src, double* dst, int doubles) {
  for (int j = threadIdx.x; j < doubles; j += kBlock) dst[j] = src[j];
}
#else
constexpr bool kDefectLoop = false;
#endif

// One ray per thread (grid_stream), no loop.  Slots beyond n fall outside every descriptor: their loads return 0
// (alive = 0) and their stores are dropped, so the tail needs no branch.
template <int KIND, bool DEFECT>
__global__ __launch_bounds__(kBlock) void k_trace_element(const ElemArg ea, const ArtBundleView in,
                                                          const ArtBundleView out, const int64_t n, const int xmap) {
  const ArtElementDesc& e = ea.e[blockIdx.y];
  const double* zern = e.zern;
#ifdef ART_ZERN_LDS
  __shared__ double s_zern[DEFECT ? 4 * ART_ZERN_STRIDE : 2];   // (comparison build: at most 4 tables per element)
  if (DEFECT) {
    stage_tables(e.zern, s_zern, e.n_defects * ART_ZERN_STRIDE);
    __syncthreads();
    zern = s_zern;
  }
#endif
  const BundleRsrc bi = make_rsrc(in, n), bo = make_rsrc(out, n);
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  int64_t i = tile_of(blockIdx.x, gridDim.x, xmap) * kBlock + threadIdx.x;
  do {
    art::Ray r;
    r.inc = 0.0;
    uint8_t a;
    load_slot(bi, i, r, a);
    bool ok = a != 0;
#ifdef ART_DIAG_NOCOMPUTE   // timing-only build: memory traffic without the intersection math (results are wrong)
    r.path += e.mp[0];
#else
    if (ok) ok = art::trace_ray<KIND, DEFECT>(e, zern, r);
#endif
    store_slot(bo, i, r, ok);
    i += stride;
  } while (DEFECT && kDefectLoop && i < n);
}

// Elements whose Zernike tables are in the recurrence layout (ART_FLAG_ZERN_RECURRENCE: orders above 16): the
// reference's recurrences per ray, three rotating rows of values and derivatives in per-lane arrays (private memory).
// A kernel of its own, so that the register-resident kernels above stay free of scratch; any optic kind (run-time
// switch), one ray per thread.
__global__ __launch_bounds__(kBlock) void k_trace_element_zrec(const ElemArg ea, const ArtBundleView in,
                                                               const ArtBundleView out, const int64_t n) {
  const ArtElementDesc& e = ea.e[blockIdx.y];
  const BundleRsrc bi = make_rsrc(in, n), bo = make_rsrc(out, n);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  art::Ray r;
  r.inc = 0.0;
  uint8_t a;
  load_slot(bi, i, r, a);
  bool ok = a != 0;
  if (ok) ok = art::trace_ray<ART_KIND_DYN, true, true>(e, e.zern, r);
  store_slot(bo, i, r, ok);
}

using art::ChainArgs;
using art::kChainMax;

#ifndef ART_STORE_DIRECT
// The 8 fp64 outputs of an element go through LDS so that every wave instruction stores 16 B per lane = 1 KiB of ONE
// stream (wave w owns streams 2w and 2w+1 of the workgroup's 256 rays) instead of 8 B per lane = 512 B: half the store
// instructions per element.  Mid-round, with the kernel bound by arithmetic, this bought nothing; with a quarter of the
// instructions gone it is 4-5 % on relay4 and 11 % on the 8-element C4 chain (tools/r02_exp22.sh; -DART_STORE_DIRECT is
// the 8-byte form).  A PAIR of slots is written when either ray is alive (the dead one's slot receives its last live
// state: unspecified by contract, its alive byte is 0); pairs of dead rays are dropped by the range check, so a masked
// bundle costs the bytes of its survivors, as before.  The two barriers order LDS only (no vmcnt wait: ISA checked).
__device__ __forceinline__ void store_tile_lds4(const ArtBundleView& v, const int64_t n, const int64_t first,
                                                const int64_t tile0, const unsigned t, const art::Ray& r, const bool ok,
                                                double (*s_out)[kBlock], uint8_t* s_al) {
  s_out[0][t] = r.ox; s_out[1][t] = r.oy; s_out[2][t] = r.oz;
  s_out[3][t] = r.dx; s_out[4][t] = r.dy; s_out[5][t] = r.dz;
  s_out[6][t] = r.path; s_out[7][t] = r.inc;
  s_al[t] = (uint8_t)(ok ? 1 : 0);
  const unsigned nb = (unsigned)(v.alive != nullptr ? n : 0);
  __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(ok ? 1 : 0), rsrc_of(v.alive + first, nb), (int)(unsigned)(tile0 + t), 0,
                                       ART_ST_AUX);
  __syncthreads();
  const int w = __builtin_amdgcn_readfirstlane((int)(t >> 6)), l = (int)(t & 63);
  double* const* rows = &v.ox;   // ox, oy, oz, dx, dy, dz, path, incidence are consecutive pointers
  bool keep[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) keep[h] = (s_al[h * 128 + 2 * l] | s_al[h * 128 + 2 * l + 1]) != 0;
#pragma unroll
  for (int js = 0; js < 2; ++js) {
    const int j = 2 * w + js;
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(rows[j] + first, nb * 8u);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int s0 = h * 128 + 2 * l;
      st_2f64(rs, keep[h] ? (unsigned)(tile0 + s0) * 8u : kDropOffset, s_out[j][s0], s_out[j][s0 + 1]);
    }
  }
  __syncthreads();
}
#endif

// Whole chain with the ray resident in registers; history written for every element whose view is non-null.
// `a` lives in kernel arguments (k_trace_chain) or in the device-resident scene table (k_trace_scene): either way
// its fields are wave-uniform and fetched by scalar loads where they are used.  Rays [first, first + n) of every view.
// `bx` / `nbx`: this workgroup's index among the launch's tile workgroups and their number -- blockIdx.x / gridDim.x in the
// tile-major launches, blockIdx.y / gridDim.y in the chain-interleaved scene launch (k_trace_scene, `transposed`).
// KIND1 >= 0: a chain of ONE element whose optic kind is known at compile time (the one-element defect chains: C5) -- no
// run-time switch on the kind, no element loop; ART_KIND_DYN: the general body.
template <bool DEFECT, int KIND1 = ART_KIND_DYN>
__device__ __forceinline__ void chain_body(const ChainArgs& a, const int64_t first, const int64_t n, const int xmap,
                                           double* s_zern, const unsigned bx, const unsigned nbx, const bool keep_in = false) {
#ifdef ART_ZERN_LDS
  if (DEFECT) {
    int off = 0;
    for (int k = 0; k < a.n_elems; ++k) {
      stage_tables(a.e[k].zern, s_zern + off, a.e[k].n_defects * ART_ZERN_STRIDE);
      off += a.e[k].n_defects * ART_ZERN_STRIDE;
    }
    __syncthreads();
  }
#endif
  __shared__ double s_w[kBlock];   // per-lane parking slot (no barrier: a lane only reads what it wrote)
  __shared__ __attribute__((aligned(16))) double s_red[(kBlock / 64) * 8 * kTileStride];   // wave-private tiles of wave_reduce24 (18 KiB)
#ifndef ART_STORE_DIRECT
  // the staging tile of the 16-byte stores shares the tail's reduction tiles: the tail runs behind the last element's
  // closing barrier
  double (*s_out)[kBlock] = reinterpret_cast<double (*)[kBlock]>(s_red);
  static_assert(sizeof(double) * 8 * kBlock <= sizeof(double) * (kBlock / 64) * 8 * kTileStride, "staging tile fits");
  __shared__ uint8_t s_al[kBlock];
#endif
  __shared__ double s_part[kBlock / 64][kReadoutSlots];   // wave totals of the fused read-out, combined per workgroup
  const int64_t tile = tile_of(bx, nbx, xmap);
  const BundleRsrc bi = make_rsrc(a.in, n, first);
  const int64_t stride = (int64_t)nbx * kBlock;
  int64_t i0 = tile * kBlock + threadIdx.x;
  do {
    // One register identifies the lane from here on: the slot index, made opaque so that the compiler derives both the
    // buffer offsets and the lane's position in its workgroup (tile * 256 is a multiple of 256) from IT instead of
    // keeping threadIdx.x alive next to it -- at 80 registers (6 waves per SIMD) that second copy is spilled and its
    // reload in the tail is a VMEM load behind 32 stores.
    unsigned slot = (unsigned)i0;
    asm volatile("" : "+v"(slot));
    const int64_t i = slot;
    const unsigned lane = slot & (kBlock - 1);
    art::Ray r;
    r.inc = 0.0;
    uint8_t al;
    if (keep_in) load_slot_keep(bi, i, r, al);      // (wave-uniform: the launch's grid shape)
    else load_slot(bi, i, r, al);
    // Weight of the fused read-out: fetched with the ray (a zero-length descriptor without read-out or weights) and
    // parked in LDS until the tail needs it -- held in registers across the chain it costs the 2 VGPRs that push the
    // 5-wave build into scratch, and a scratch reload at the end is a VMEM load behind 36 stores (see the tail).
    s_w[lane] = ld_f64(rsrc_of(const_cast<double*>(a.ro.w) + first,
                                      ((a.flags & art::kFlagReadout) && a.ro.w && !a.ro.lite) ? (unsigned)(n * 8) : 0u),
                       (unsigned)i * 8u);
    bool ok = al != 0;
    const double* zk = s_zern;
    // do-while (n_elems >= 1, checked on the host): with a zero-trip path the ray loads above would still be in
    // flight where that path joins the tail, and the ONE wait the compiler then places at the join makes the main
    // path wait for the acknowledgement of all its stores (vmcnt counts loads and stores in one in-order queue)
    int k = 0;
    do {
#ifdef ART_DIAG_NOCOMPUTE
      r.path += a.e[k].mp[0];
#else
#ifdef ART_ZERN_LDS
      if (ok) ok = art::trace_ray_dyn<DEFECT>(a.e[k], zk, r);
      zk += a.e[k].n_defects * ART_ZERN_STRIDE;
#else
      if (KIND1 != ART_KIND_DYN) {
        if (ok) ok = art::trace_ray<(KIND1 < 0 ? 0 : KIND1), DEFECT>(a.e[0], a.e[0].zern, r);
      } else {
        if (ok) ok = art::trace_ray_dyn<DEFECT>(a.e[k], a.e[k].zern, r);
      }
#endif
#endif
#ifndef ART_STORE_DIRECT
      store_tile_lds4(a.out[k], n, first, (int64_t)(slot - lane), lane, r, ok, s_out, s_al);
#else
      // no history view for this element -> zero-length descriptors: every store is dropped by the range check
      store_slot(make_rsrc(a.out[k], a.out[k].alive != nullptr ? n : 0, first), i, r, ok);
#endif
    } while (KIND1 == ART_KIND_DYN && ++k < a.n_elems);
    if (a.flags & art::kFlagReadout) {
      // Fused detector read-out of the last bundle (art_trace_chain_readout): the ray is still in registers.  X, Y, opl
      // of dead rays are dropped by the range check.  Statistics: per wave wave_reduce24, per workgroup one partial,
      // written row by row (row_of_slot) into ro.scratch -- no __syncthreads() and no load down here: either would make
      // every wave wait for the acknowledgement of its 36 outstanding stores (vmcnt counts loads and stores in one queue)
      // instead of retiring as soon as they are issued, which cost 35 %.
      if (a.ro.sums) {     // (wave-uniform) pass (1) of the analysis instead of a read-out: see run_sums8
        double v[8];
        sums_values(v, ok, r, a.ro.w ? s_w[lane ^ ((unsigned)a.flags >> 30)] : 1.0);
        const double tot = run_sums8(v, s_red + (lane >> 6) * (8 * kTileStride), lane & 63);
        tile_sums_store(tot, __ballot(ok), &s_part[0][0], lane, a.ro.scratch, (n + kBlock - 1) / kBlock, tile);
      } else {
      double acc[kReadoutSlots];
      double Ix, Iy, Iz, x = 0.0, y = 0.0, o = 0.0;
      if (ok) art::detector_ray(a.ro.det, r, Ix, Iy, Iz, x, y, o);
      const unsigned nb8 = (unsigned)(n * 8);
      const unsigned o8 = ok ? (unsigned)i * 8u : kDropOffset;
#ifndef ART_DIAG_RO_NOXYO    // timing-only builds of the fused tail: without its per-ray outputs, ...
      st_f64(rsrc_of(a.ro.X + first, a.ro.X ? nb8 : 0u), o8, x);
      st_f64(rsrc_of(a.ro.Y + first, a.ro.Y ? nb8 : 0u), o8, y);
      st_f64(rsrc_of(a.ro.opl + first, a.ro.opl ? nb8 : 0u), o8, o);
#endif
      // re-read the parked weight through an index the compiler cannot prove equal to the one it was stored with
      // (flags has no bit 30), so that the value is neither kept in a register nor spilled
      double tot[3];
      if (a.ro.lite) {     // (wave-uniform: the flag sits in the descriptor)
        readout_single_lite(acc, ok, x, y, o);
        wave_reduce_lite(acc, s_red + (lane >> 6) * (8 * kTileStride), lane & 63, tot);
      } else {
      readout_single(acc, ok, x, y, o, s_w[lane ^ ((unsigned)a.flags >> 30)], a.ro.w != nullptr, a.ro.cx, a.ro.cy,
                     a.ro.co);
#ifdef ART_DIAG_RO_NOREDUCE  // ... without the wave reduction, ...
      tot[0] = acc[0] + acc[1] + acc[6] + acc[7] + acc[8] + acc[9] + acc[10] + acc[11];
      tot[1] = acc[16] + acc[17] + acc[18] + acc[19] + acc[20] + acc[21];
      tot[2] = acc[2] + acc[3] + acc[4] + acc[5] + acc[12] + acc[13];
#else
      wave_reduce24(acc, s_red + (lane >> 6) * (8 * kTileStride), lane & 63, tot);
#endif
      }
      // One partial per WORKGROUP (round 3; per wave before): the four wave totals meet in LDS behind a bare s_barrier --
      // __syncthreads() would also wait for the acknowledgement of the stores in flight -- and threads 0..21 fold them in
      // wave order and store row `thread` (= pass * 8 + stat, row_of_slot) of the scratch area.  A quarter of the partials:
      // the fold that follows reads 7.5 MB instead of 30 MB per 1e7 rays and fits ONE launch (launch_fold_*).
      if ((lane & 7) == 0) {
        const int stat = (lane & 63) >> 3, w = lane >> 6;
        s_part[w][stat] = tot[0];
        s_part[w][8 + stat] = tot[1];
        if (stat < 6) s_part[w][16 + stat] = tot[2];
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef ART_DIAG_RO_NOSCRATCH  // ... without the partial-statistics stores
      if (tot[0] + tot[1] + tot[2] == -1.2345e300) {
#else
      if (lane < kUsedSlots) {
#endif
        double v = s_part[0][lane];
#pragma unroll
        for (int j = 1; j < kBlock / 64; ++j) {
          const double q = s_part[j][lane];
          v = (lane < 16) ? v + q : (lane < 19 ? fmin(v, q) : fmax(v, q));    // rows 16-18: minima, 19-21: maxima
        }
        a.ro.scratch[(int64_t)lane * nbx + bx] = v;
      }
      }
    }
    i0 += stride;
  } while (DEFECT && kDefectLoop && i0 < n);
}

// ---- two rays per lane (round-3 experiment, ART_CHAIN_RPL=2) -----------------------------------------------------------
// A lane owns the two NEIGHBOURING slots 2p, 2p + 1: every stream is then read and written with 16 bytes per lane (1 KiB
// per wave instruction) straight from registers -- no LDS staging tile, no workgroup barrier between the elements (the
// one-ray-per-lane body needs two per element to regroup its 8-byte values into 16-byte stores), per-row descriptors and
// hardware range checks as before (a pair of dead rays is dropped by its offset, the odd tail by the per-dword check).
// The two rays of a lane are traced one after the other; a wave carries 128 rays.
template <bool DEFECT>
__device__ __forceinline__ void chain_body2(const ChainArgs& a, const int64_t first, const int64_t n, const int xmap,
                                            const unsigned bx, const unsigned nbx, const bool keep_in = false) {
  __shared__ double s_w[2][kBlock];   // per-lane parking slots of the two weights (no barrier: a lane reads what it wrote)
  __shared__ __attribute__((aligned(16))) double s_red[(kBlock / 64) * 8 * kTileStride];   // wave-private tiles of wave_reduce24
  __shared__ double s_part[kBlock / 64][kReadoutSlots];
  const int64_t tile = tile_of(bx, nbx, xmap);
  const BundleRsrc bi = make_rsrc(a.in, n, first);
  unsigned pair = (unsigned)(tile * kBlock + threadIdx.x);
  asm volatile("" : "+v"(pair));                      // ONE register identifies the lane (see chain_body)
  const unsigned lane = pair & (kBlock - 1);
  const unsigned o16 = pair * 16u, o2 = pair * 2u;
  art::Ray r[2];
  bool ok[2];
  {
    D2 ox, oy, oz, dx, dy, dz, pa;
    if (keep_in) {       // the shared input of a chain-interleaved scene launch: default cache policy (see ld_f64_keep)
      ox = ld_2f64_keep(bi.ox, o16); oy = ld_2f64_keep(bi.oy, o16); oz = ld_2f64_keep(bi.oz, o16);
      dx = ld_2f64_keep(bi.dx, o16); dy = ld_2f64_keep(bi.dy, o16); dz = ld_2f64_keep(bi.dz, o16);
      pa = ld_2f64_keep(bi.path, o16);
    } else {
      ox = ld_2f64(bi.ox, o16); oy = ld_2f64(bi.oy, o16); oz = ld_2f64(bi.oz, o16);
      dx = ld_2f64(bi.dx, o16); dy = ld_2f64(bi.dy, o16); dz = ld_2f64(bi.dz, o16);
      pa = ld_2f64(bi.path, o16);
    }
    // The two alive bytes with ONE 16-bit load.  A 16-bit access straddling the end of an odd-length array is dropped as a
    // whole: for odd n (wave-uniform) the lane that holds the last slot fetches its byte separately.
    unsigned al2 = __builtin_amdgcn_raw_buffer_load_b16(bi.alive, (int)o2, 0, ART_LD_AUX);
    if (n & 1) al2 |= __builtin_amdgcn_raw_buffer_load_b8(bi.alive, (o2 + 1u == (unsigned)n) ? (int)o2 : (int)kDropOffset, 0, ART_LD_AUX);
    ok[0] = (al2 & 0xffu) != 0;
    ok[1] = (al2 & 0xff00u) != 0;
    const D2 wv = ld_2f64(rsrc_of(const_cast<double*>(a.ro.w) + first,
                                  ((a.flags & art::kFlagReadout) && a.ro.w && !a.ro.lite) ? (unsigned)(n * 8) : 0u), o16);
    s_w[0][lane] = wv.a; s_w[1][lane] = wv.b;
    r[0].ox = ox.a; r[0].oy = oy.a; r[0].oz = oz.a; r[0].dx = dx.a; r[0].dy = dy.a; r[0].dz = dz.a; r[0].path = pa.a;
    r[1].ox = ox.b; r[1].oy = oy.b; r[1].oz = oz.b; r[1].dx = dx.b; r[1].dy = dy.b; r[1].dz = dz.b; r[1].path = pa.b;
    r[0].inc = 0.0; r[1].inc = 0.0;
  }
  int k = 0;
  do {
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (ok[h]) ok[h] = art::trace_ray_dyn<DEFECT>(a.e[k], a.e[k].zern, r[h]);
    const ArtBundleView& v = a.out[k];
    const unsigned nb = (unsigned)(v.alive != nullptr ? n : 0);     // no history view: every store is dropped
    const unsigned off = (ok[0] || ok[1]) ? o16 : kDropOffset;
    st_2f64(rsrc_of(v.ox + first, nb * 8u), off, r[0].ox, r[1].ox);
    st_2f64(rsrc_of(v.oy + first, nb * 8u), off, r[0].oy, r[1].oy);
    st_2f64(rsrc_of(v.oz + first, nb * 8u), off, r[0].oz, r[1].oz);
    st_2f64(rsrc_of(v.dx + first, nb * 8u), off, r[0].dx, r[1].dx);
    st_2f64(rsrc_of(v.dy + first, nb * 8u), off, r[0].dy, r[1].dy);
    st_2f64(rsrc_of(v.dz + first, nb * 8u), off, r[0].dz, r[1].dz);
    st_2f64(rsrc_of(v.path + first, nb * 8u), off, r[0].path, r[1].path);
    st_2f64(rsrc_of(v.incidence + first, nb * 8u), off, r[0].inc, r[1].inc);
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(v.alive + first, nb);
    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)((ok[0] ? 1u : 0u) | (ok[1] ? 0x100u : 0u)), ra, (int)o2, 0, ART_ST_AUX);
    if (n & 1)      // odd n: the 16-bit store of the pair that holds the last slot was dropped by the range check
      __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(ok[0] ? 1 : 0), ra, (o2 + 1u == (unsigned)n) ? (int)o2 : (int)kDropOffset, 0, ART_ST_AUX);
  } while (++k < a.n_elems);
  if ((a.flags & art::kFlagReadout) && a.ro.sums) {     // pass (1) of the analysis instead of a read-out: see run_sums8
    const unsigned li = lane ^ ((unsigned)a.flags >> 30);
    double v0[8], v1[8], totA, totB;
    sums_values(v0, ok[0], r[0], a.ro.w ? s_w[0][li] : 1.0);
    sums_values(v1, ok[1], r[1], a.ro.w ? s_w[1][li] : 1.0);
    run_sums8_pairs(v0, v1, s_red + (lane >> 6) * (8 * kTileStride), lane & 63, totA, totB);
    static_assert(sizeof(s_part) >= sizeof(double) * 8 * kSumRows, "eight runs fit the partial area");
    tile_sums_store_pairs(totA, totB, __ballot(ok[0]), __ballot(ok[1]), &s_part[0][0], lane, a.ro.scratch,
                          (n + kBlock - 1) / kBlock, tile);
  } else if (a.flags & art::kFlagReadout) {
    double acc[kReadoutSlots];
    double Ix, Iy, Iz, x[2] = {0.0, 0.0}, y[2] = {0.0, 0.0}, o[2] = {0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (ok[h]) art::detector_ray(a.ro.det, r[h], Ix, Iy, Iz, x[h], y[h], o[h]);
    const unsigned nb8 = (unsigned)(n * 8);
    const unsigned off = (ok[0] || ok[1]) ? o16 : kDropOffset;
    st_2f64(rsrc_of(a.ro.X + first, a.ro.X ? nb8 : 0u), off, x[0], x[1]);
    st_2f64(rsrc_of(a.ro.Y + first, a.ro.Y ? nb8 : 0u), off, y[0], y[1]);
    st_2f64(rsrc_of(a.ro.opl + first, a.ro.opl ? nb8 : 0u), off, o[0], o[1]);
    const unsigned li = lane ^ ((unsigned)a.flags >> 30);     // an index the compiler cannot prove equal to the parking one
    double tot[3];
    if (a.ro.lite) {
      readout_single_lite(acc, ok[0], x[0], y[0], o[0]);
      readout_accumulate_lite(acc, ok[1], x[1], y[1], o[1]);
      wave_reduce_lite(acc, s_red + (lane >> 6) * (8 * kTileStride), lane & 63, tot);
    } else {
      readout_single(acc, ok[0], x[0], y[0], o[0], s_w[0][li], a.ro.w != nullptr, a.ro.cx, a.ro.cy, a.ro.co);
      readout_accumulate(acc, ok[1], x[1], y[1], o[1], s_w[1][li], a.ro.w != nullptr, a.ro.cx, a.ro.cy, a.ro.co);
      wave_reduce24(acc, s_red + (lane >> 6) * (8 * kTileStride), lane & 63, tot);
    }
    if ((lane & 7) == 0) {
      const int stat = (lane & 63) >> 3, w = lane >> 6;
      s_part[w][stat] = tot[0];
      s_part[w][8 + stat] = tot[1];
      if (stat < 6) s_part[w][16 + stat] = tot[2];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (lane < kUsedSlots) {
      double t = s_part[0][lane];
#pragma unroll
      for (int j = 1; j < kBlock / 64; ++j) {
        const double q = s_part[j][lane];
        t = (lane < 16) ? t + q : (lane < 19 ? fmin(t, q) : fmax(t, q));
      }
      a.ro.scratch[(int64_t)lane * nbx + bx] = t;
    }
  }
}

// Fold of the fused read-out's per-WORKGROUP partials (row-major: scratch[row_of_slot(slot) * nparts + part], 39 063
// parts per 1e7 rays), always in a fixed order (deterministic).  ONE launch up to kFoldDirect partials (1.67e7 rays):
// a 1024-thread workgroup per statistic folds its whole row (312 KB at 1e7 rays: thread t takes parts t, t + 1024, ...,
// ~40 independent loads, then a shuffle tree and 16 wave totals through LDS).  Round 2 left one partial per WAVE and
// folded the four-times-longer rows in two launches (64 chunks per statistic, then one workgroup per statistic): 8 + 5 us
// behind every relay4 trace, 18 us behind C2's 11 chains; that two-stage form remains for longer rows.
// Grids: (24, 1, chains) direct; (24, chunks, chains) + (24, 1, chains) beyond.
constexpr int kFoldChunks = 64;
constexpr int kFoldBlock = 1024;
constexpr int64_t kFoldDirect = 65536;
__device__ __forceinline__ double fold_op(const double a, const double b, const int op_k) {
  return (op_k == RSUM) ? a + b : (op_k == RMIN ? fmin(a, b) : fmax(a, b));
}
// every thread of the workgroup (blockDim.x threads, a multiple of 64) calls this; thread 0 stores the result
__device__ __forceinline__ void fold_range(const double* row, const int64_t lo, const int64_t hi, const int op_k,
                                           double* out) {
  __shared__ double s[kFoldBlock / 64];
  const double ident = (op_k == RSUM) ? 0.0 : (op_k == RMIN ? INFINITY : -INFINITY);
  double acc = ident;
  // eight loads in flight per thread before the first is folded (a thread's ~40 values one load at a time are 40 memory
  // latencies in a row: 17 us per 1e7 rays, measured); the fold order stays fixed: p, p + B, p + 2B, ...
  constexpr int kU = 8;
  for (int64_t p = lo + threadIdx.x; p < hi; p += (int64_t)blockDim.x * kU) {
    double v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t q = p + (int64_t)u * blockDim.x;
      v[u] = (q < hi) ? row[q] : ident;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) acc = fold_op(acc, v[u], op_k);
  }
  acc = wave_reduce(acc, op_k);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = s[0];
    for (int j = 1; j < (int)(blockDim.x >> 6); ++j) v = fold_op(v, s[j], op_k);
    *out = v;
  }
}
__device__ __forceinline__ double* fold_mid(double* scratch, const int64_t nparts) {
  return scratch + (int64_t)kReadoutSlots * nparts;
}
__device__ __forceinline__ int fold_row_of(const int slot) {
  int row = 0;
#pragma unroll
  for (int k = 0; k < kUsedSlots; ++k) row = (slot == k) ? row_of_slot(k) : row;
  return row;
}
__device__ __forceinline__ void fold_stage1(double* scratch, const int64_t nparts) {
  if (blockIdx.x >= kUsedSlots) return;
  const int ops[kReadoutSlots] = ART_READOUT_OPS;
  const int64_t per = (nparts + kFoldChunks - 1) / kFoldChunks;
  const int64_t lo = (int64_t)blockIdx.y * per, hi = (lo + per < nparts) ? lo + per : nparts;
  fold_range(scratch + (int64_t)fold_row_of(blockIdx.x) * nparts, lo, hi, ops[blockIdx.x],
             fold_mid(scratch, nparts) + blockIdx.x * kFoldChunks + blockIdx.y);
}
__device__ __forceinline__ void fold_stage2(double* scratch, const int64_t nparts, double* out24, const int direct) {
  if (blockIdx.x >= kUsedSlots) {
    if (threadIdx.x == 0) out24[blockIdx.x] = 0.0;
    return;
  }
  const int ops[kReadoutSlots] = ART_READOUT_OPS;
  if (direct) {   // nparts == 0 (an empty bundle): the fold of nothing leaves the identities
    fold_range(scratch + (int64_t)fold_row_of(blockIdx.x) * nparts, 0, nparts, ops[blockIdx.x], out24 + blockIdx.x);
    return;
  }
  fold_range(fold_mid(scratch, nparts) + blockIdx.x * kFoldChunks, 0, kFoldChunks, ops[blockIdx.x], out24 + blockIdx.x);
}
__global__ __launch_bounds__(kFoldBlock) void k_chain_readout_fold1(const ChainArgs* __restrict__ tab, const int64_t nparts) {
  fold_stage1(tab[blockIdx.z].ro.scratch, nparts);
}
__global__ __launch_bounds__(kFoldBlock) void k_chain_readout_fold2(const ChainArgs* __restrict__ tab, const int64_t nparts,
                                                                    const int direct) {
  fold_stage2(tab[blockIdx.z].ro.scratch, nparts, tab[blockIdx.z].ro.out24, direct);
}
__global__ __launch_bounds__(kFoldBlock) void k_chain_readout_fold1_one(double* scratch, const int64_t nparts) {
  fold_stage1(scratch, nparts);
}
__global__ __launch_bounds__(kFoldBlock) void k_chain_readout_fold2_one(double* scratch, double* out24, const int64_t nparts,
                                                                        const int direct) {
  fold_stage2(scratch, nparts, out24, direct);
}

// launch the fold (host side): ONE launch up to kFoldDirect partials; a short row does not need 1024 threads
inline int fold_threads(int64_t nparts) { return nparts <= 4096 ? kBlock : kFoldBlock; }
inline void launch_fold_one(double* scratch, double* out24, int64_t nparts, hipStream_t s) {
  const int direct = nparts <= kFoldDirect;
  if (!direct)
    hipLaunchKernelGGL(k_chain_readout_fold1_one, dim3(kReadoutSlots, kFoldChunks), dim3(kFoldBlock), 0, s, scratch, nparts);
  hipLaunchKernelGGL(k_chain_readout_fold2_one, dim3(kReadoutSlots), dim3(direct ? fold_threads(nparts) : kBlock), 0, s, scratch,
                     out24, nparts, direct);
}
inline void launch_fold_scene(const ChainArgs* seg, int n_chains, int64_t nparts, hipStream_t s) {
  const int direct = nparts <= kFoldDirect;
  if (!direct)
    hipLaunchKernelGGL(k_chain_readout_fold1, dim3(kReadoutSlots, kFoldChunks, n_chains), dim3(kFoldBlock), 0, s, seg, nparts);
  hipLaunchKernelGGL(k_chain_readout_fold2, dim3(kReadoutSlots, 1, n_chains), dim3(direct ? fold_threads(nparts) : kBlock), 0, s,
                     seg, nparts, direct);
}

// Fold of the per-tile partial SUMS (kSumRows rows of `ntiles`, see run_sums8) into out[0 .. 8]: the same launch shape and
// fold_range for the tail of a tracing launch and for art_analyse_bundles' own pass (1) -- same bits either way.  Beyond
// kFoldDirect tiles two stages, with the chunk totals behind the rows (rows + kSumRows * ntiles, kSumRows x kFoldChunks).
__device__ __forceinline__ void sums_fold1(double* rows, const int64_t ntiles) {
  const int64_t per = (ntiles + kFoldChunks - 1) / kFoldChunks;
  const int64_t lo = (int64_t)blockIdx.y * per, hi = (lo + per < ntiles) ? lo + per : ntiles;
  fold_range(rows + (int64_t)blockIdx.x * ntiles, lo < hi ? lo : hi, hi, RSUM,
             rows + (int64_t)kSumRows * ntiles + blockIdx.x * kFoldChunks + blockIdx.y);
}
__device__ __forceinline__ void sums_fold2(double* rows, const int64_t ntiles, double* out, const int direct) {
  if (direct) fold_range(rows + (int64_t)blockIdx.x * ntiles, 0, ntiles, RSUM, out + blockIdx.x);
  else fold_range(rows + (int64_t)kSumRows * ntiles + blockIdx.x * kFoldChunks, 0, kFoldChunks, RSUM, out + blockIdx.x);
}
__global__ __launch_bounds__(kFoldBlock) void k_chain_sums_fold1(const ChainArgs* __restrict__ tab, const int64_t ntiles) {
  sums_fold1(tab[blockIdx.z].ro.scratch, ntiles);
}
__global__ __launch_bounds__(kFoldBlock) void k_chain_sums_fold2(const ChainArgs* __restrict__ tab, const int64_t ntiles,
                                                                 const int direct) {
  sums_fold2(tab[blockIdx.z].ro.scratch, ntiles, tab[blockIdx.z].ro.out24, direct);
  if (blockIdx.x == 0 && threadIdx.x < kReadoutSlots - kSumRows) tab[blockIdx.z].ro.out24[kSumRows + threadIdx.x] = 0.0;
}
__global__ __launch_bounds__(kFoldBlock) void k_chain_sums_fold1_one(double* scratch, const int64_t ntiles) {
  sums_fold1(scratch, ntiles);
}
__global__ __launch_bounds__(kFoldBlock) void k_chain_sums_fold2_one(double* scratch, double* out24, const int64_t ntiles,
                                                                     const int direct) {
  sums_fold2(scratch, ntiles, out24, direct);
  if (blockIdx.x == 0 && threadIdx.x < kReadoutSlots - kSumRows) out24[kSumRows + threadIdx.x] = 0.0;
}
inline void launch_sums_fold_one(double* scratch, double* out24, int64_t ntiles, hipStream_t s) {
  const int direct = ntiles <= kFoldDirect;
  if (!direct)
    hipLaunchKernelGGL(k_chain_sums_fold1_one, dim3(kSumRows, kFoldChunks), dim3(kFoldBlock), 0, s, scratch, ntiles);
  hipLaunchKernelGGL(k_chain_sums_fold2_one, dim3(kSumRows), dim3(direct ? fold_threads(ntiles) : kBlock), 0, s, scratch, out24,
                     ntiles, direct);
}
inline void launch_sums_fold_scene(const ChainArgs* seg, int n_chains, int64_t ntiles, hipStream_t s) {
  const int direct = ntiles <= kFoldDirect;
  if (!direct)
    hipLaunchKernelGGL(k_chain_sums_fold1, dim3(kSumRows, kFoldChunks, n_chains), dim3(kFoldBlock), 0, s, seg, ntiles);
  hipLaunchKernelGGL(k_chain_sums_fold2, dim3(kSumRows, 1, n_chains), dim3(direct ? fold_threads(ntiles) : kBlock), 0, s, seg,
                     ntiles, direct);
}

template <bool DEFECT, int WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void k_trace_chain(const ChainArgs, const int64_t n, const int xmap) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];   // used by the -DART_ZERN_LDS build only
  // The descriptors are read where the launch put them, in the kernel-argument segment (constant address space ->
  // scalar loads, first parameter = offset 0).  Through the by-value parameter itself that is only what the compiler
  // usually makes of it: the element index is a run-time value, and one build of the defect kernel came out with the
  // whole 3.3 KB copied to scratch at entry and 452 vector loads from there.
  typedef const ChainArgs __attribute__((address_space(4)))* kernarg_t;
  chain_body<DEFECT>(*(const ChainArgs*)(kernarg_t)__builtin_amdgcn_kernarg_segment_ptr(), 0, n, xmap, s_dyn, blockIdx.x,
                     gridDim.x);
}

template <bool DEFECT, int WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void k_trace_chain2(const ChainArgs, const int64_t n, const int xmap) {
  typedef const ChainArgs __attribute__((address_space(4)))* kernarg_t;
  chain_body2<DEFECT>(*(const ChainArgs*)(kernarg_t)__builtin_amdgcn_kernarg_segment_ptr(), 0, n, xmap, blockIdx.x, gridDim.x);
}
// Which (chain, tile) a workgroup of a scene launch works on.  `shape`: 0 = tile-major grid (tiles, chains); bit 0 =
// chain-interleaved grid (chains, tiles); bit 2 = XCD-GROUPED: a 1-D grid of chains x tiles workgroups (tiles a multiple of
// 8, chain count in bits 8 and up) in which the C workgroups of one tile all run on ONE XCD.  The dispatcher hands
// consecutive workgroup ids to the 8 XCDs in turn (id mod 8) and every XCD has its own L2: in the plain interleaved grid
// the C workgroups that read the same input tile land on 8 different XCDs, so the tile crosses the fabric 8 times (from
// the memory-side cache after the first) and the L2s share nothing.  Here id = 8 k + xcd; inside an XCD consecutive k
// run through the chains of one tile (chain = k mod C), then the next tile of that XCD (tile = 8 (k / C) + xcd): the
// tile is fetched into that XCD's L2 once and the other C - 1 workgroups, dispatched right behind, hit it there.
// (bit 1 = the shared input is loaded with the default cache policy, see scene_keep.)
// The grid holds C x (tiles rounded up to a multiple of 8) workgroups; the launch's TRUE tile count arrives in the `xmap`
// parameter (the tile mapping of tile_of does not apply to this shape): workgroups of the padding leave at once, and the
// per-tile partials of a read-out keep the layout and fold order of the other shapes (same bits whatever the shape).
struct SceneWG { unsigned chain, bx, nbx; int xmap; bool idle; };
__device__ __forceinline__ SceneWG scene_wg(const int shape, const int xmap) {
  if (shape & 4) {
    const unsigned C = (unsigned)shape >> 8, id = blockIdx.x;
    const unsigned k = id >> 3, tq = k / C, bx = tq * 8u + (id & 7u);
    return {k - tq * C, bx, (unsigned)xmap, 0, bx >= (unsigned)xmap};
  }
  if (shape & 1) return {blockIdx.x, blockIdx.y, gridDim.y, xmap, false};
  return {blockIdx.y, blockIdx.x, gridDim.x, xmap, false};
}

template <bool DEFECT, int WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void k_trace_scene2(const ChainArgs* __restrict__ tab, const int64_t first,
                                                                const int64_t n, const int xmap, const int transposed) {
  const SceneWG w = scene_wg(transposed, xmap);
  if (w.idle) return;
  chain_body2<DEFECT>(tab[w.chain], first, n, w.xmap, w.bx, w.nbx, (transposed & 2) != 0);
}

// Many chains in one launch, descriptors in the device-resident scene table (art_scene.h).  Three grid shapes (scene_wg):
//   tile-major    grid (tiles, chains): blockIdx.y = chain -- the dispatcher works through one chain's tiles after the other;
//   interleaved   grid (chains, tiles): blockIdx.x = chain -- the workgroups of ONE tile of ALL chains are dispatched together;
//   XCD-grouped   1-D grid: ... and on the same XCD, so that they share the tile in that XCD's L2.
// The last two are for scenes whose chains all read the SAME input bundle (a loop list traced from one source or from its
// shared prefix, art_scene.h kFlagSharedIn): interleaved, the tile of the input that chain 0 fetches from HBM is still in the
// memory-side cache when chains 1 .. C-1 ask for it a few microseconds later, instead of being streamed from HBM once per
// chain; XCD-grouped (the default), it does not even leave the L2.
template <bool DEFECT, int WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void k_trace_scene(const ChainArgs* __restrict__ tab, const int64_t first,
                                                               const int64_t n, const int xmap, const int transposed) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  const SceneWG w = scene_wg(transposed, xmap);   // (ONE copy of the body)
  if (w.idle) return;
  chain_body<DEFECT>(tab[w.chain], first, n, w.xmap, s_dyn, w.bx, w.nbx, (transposed & 2) != 0);
}

// One-element chains WITH defects, the optic's kind a template parameter (round 5, C5: a deformed parabola + read-out).
template <int KIND1, int WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void k_trace_scene1(const ChainArgs* __restrict__ tab, const int64_t first,
                                                                const int64_t n, const int xmap, const int transposed) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  const SceneWG w = scene_wg(transposed, xmap);
  if (w.idle) return;
  chain_body<true, KIND1>(tab[w.chain], first, n, w.xmap, s_dyn, w.bx, w.nbx, (transposed & 2) != 0);
}
template <int KIND1, int WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void k_trace_chain1(const ChainArgs, const int64_t n, const int xmap) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  typedef const ChainArgs __attribute__((address_space(4)))* kernarg_t;
  chain_body<true, KIND1>(*(const ChainArgs*)(kernarg_t)__builtin_amdgcn_kernarg_segment_ptr(), 0, n, xmap, s_dyn, blockIdx.x,
                          gridDim.x);
}

// One-element chains with defects whose optic is a plane, a sphere or a parabola (what a deformed mirror usually is: C5)
// run a body compiled for that kind (k_trace_*1): no run-time switch, no torus state -- 98 VGPRs instead of 118.
// ART_CHAIN_SPECIAL=0 switches it off, ART_CHAIN_SPECIAL_WAVES=5 selects the 5-wave build (96 VGPRs, one 8-byte spill per
// ray in front of the stores); read per launch: the A/B tool alternates them inside one process.
inline int special_waves() {
  const char* e = getenv("ART_CHAIN_SPECIAL");
  if (e && e[0] == '0') return 0;
  const char* w = getenv("ART_CHAIN_SPECIAL_WAVES");
  return (w && atoi(w) == 5) ? 5 : 4;
}
inline bool special_kind(const int kind) { return kind == ART_PLANE || kind == ART_SPHERE || kind == ART_PARABOLA; }
template <int WAVES>
void launch_scene1(const int kind, const dim3 g, hipStream_t s, const ChainArgs* seg, int64_t off, int64_t cnt, int xm, int tr) {
  const dim3 b(kBlock);
  if (kind == ART_PLANE) hipLaunchKernelGGL((k_trace_scene1<ART_PLANE, WAVES>), g, b, 0, s, seg, off, cnt, xm, tr);
  else if (kind == ART_SPHERE) hipLaunchKernelGGL((k_trace_scene1<ART_SPHERE, WAVES>), g, b, 0, s, seg, off, cnt, xm, tr);
  else hipLaunchKernelGGL((k_trace_scene1<ART_PARABOLA, WAVES>), g, b, 0, s, seg, off, cnt, xm, tr);
}
template <int WAVES>
void launch_chain1(const int kind, const dim3 g, hipStream_t s, const ChainArgs& a, int64_t cnt, int xm) {
  const dim3 b(kBlock);
  if (kind == ART_PLANE) hipLaunchKernelGGL((k_trace_chain1<ART_PLANE, WAVES>), g, b, 0, s, a, cnt, xm);
  else if (kind == ART_SPHERE) hipLaunchKernelGGL((k_trace_chain1<ART_SPHERE, WAVES>), g, b, 0, s, a, cnt, xm);
  else hipLaunchKernelGGL((k_trace_chain1<ART_PARABOLA, WAVES>), g, b, 0, s, a, cnt, xm);
}

// ------------------------------------------------------------------------------------------- AoS -> SoA
__global__ __launch_bounds__(kBlock) void k_pack_rays(const double* __restrict__ points, const double* __restrict__ vectors,
                                                      const double* __restrict__ path0, const int64_t n,
                                                      const ArtBundleView out) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    art::Ray r;
    r.ox = points[3 * i]; r.oy = points[3 * i + 1]; r.oz = points[3 * i + 2];
    const double vx = vectors[3 * i], vy = vectors[3 * i + 1], vz = vectors[3 * i + 2];
    const double inv = 1.0 / sqrt(art::dot3(vx, vy, vz, vx, vy, vz));
    r.dx = vx * inv; r.dy = vy * inv; r.dz = vz * inv;
    r.path = path0 ? path0[i] : 0.0;
    r.inc = NAN;
    store_ray(out, i, r);
    out.alive[i] = 1;
  }
}

// ------------------------------------------------------------------------------------------- affine map
__global__ __launch_bounds__(kBlock) void k_transform(const ArtDetectorDesc mt, const int rotate_points,
                                                      const ArtBundleView in, const ArtBundleView out, const int64_t n) {
  // mt.rot = M, mt.centre = T (ArtDetectorDesc reused as a POD carrier)
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    art::Ray r;
    load_ray(in, i, r);
    r.inc = in.incidence[i];
    const uint8_t a = in.alive[i];
    double px = r.ox, py = r.oy, pz = r.oz, vx, vy, vz;
    if (rotate_points) art::mat3_apply(mt.rot, r.ox, r.oy, r.oz, px, py, pz);
    art::mat3_apply(mt.rot, r.dx, r.dy, r.dz, vx, vy, vz);
    const double inv = 1.0 / sqrt(art::dot3(vx, vy, vz, vx, vy, vz));   // Ray.vector setter
    r.ox = px + mt.centre[0]; r.oy = py + mt.centre[1]; r.oz = pz + mt.centre[2];
    r.dx = vx * inv; r.dy = vy * inv; r.dz = vz * inv;
    store_ray(out, i, r);
    out.alive[i] = a;
  }
}

// ------------------------------------------------------------------------------------------- detector
__global__ __launch_bounds__(kBlock) void k_detector(const ArtDetectorDesc d, const ArtBundleView b, const int64_t n,
                                                     double* p3x, double* p3y, double* p3z, double* X, double* Y,
                                                     double* opl) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    if (b.alive[i] == 0) continue;
    art::Ray r;
    load_ray(b, i, r);
    double Ix, Iy, Iz, x, y, o;
    art::detector_ray(d, r, Ix, Iy, Iz, x, y, o);
    if (p3x) { p3x[i] = Ix; p3y[i] = Iy; p3z[i] = Iz; }
    if (X) { X[i] = x; Y[i] = y; }
    if (opl) opl[i] = o;
  }
}

template <bool HAS_W>
__global__ __launch_bounds__(kBlock) void k_stats_partial(const uint8_t* alive, const double* X, const double* Y,
                                                          const double* opl, const double* w, const int64_t n,
                                                          double* scratch) {
  const int ops[kRedSlots] = {RSUM, RSUM, RMIN, RMAX, RMIN, RMAX, RSUM, RSUM,
                              RSUM, RSUM, RSUM, RSUM, RMIN, RMAX, RSUM, RSUM};
  double acc[kRedSlots];
#pragma unroll
  for (int k = 0; k < kRedSlots; ++k) acc[k] = (ops[k] == RSUM) ? 0.0 : (ops[k] == RMIN ? INFINITY : -INFINITY);
  const int64_t stride = (int64_t)gridDim.x * kBlock, last = n - 1;
  for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < n; i0 += kRedUnroll * stride) {
    double xv[kRedUnroll], yv[kRedUnroll], ov[kRedUnroll], wv[kRedUnroll];
    bool live[kRedUnroll];
#pragma unroll
    for (int u = 0; u < kRedUnroll; ++u) {
      const int64_t i = i0 + u * stride, c = i < n ? i : last;
      live[u] = (ld_nt(alive + c) != 0) & (i < n);
      xv[u] = X ? ld_nt(X + c) : 0.0;      // (wave-uniform: absent streams read 0)
      yv[u] = Y ? ld_nt(Y + c) : 0.0;
      ov[u] = opl ? ld_nt(opl + c) : 0.0;
      wv[u] = HAS_W ? ld_nt(w + c) : 1.0;
    }
#pragma unroll
    for (int u = 0; u < kRedUnroll; ++u) {
      const bool l = live[u];
      const double x = l ? xv[u] : 0.0, y = l ? yv[u] : 0.0, o = l ? ov[u] : 0.0, ww = l ? wv[u] : 0.0;
      acc[0] += l ? 1.0 : 0.0; acc[1] += o;
      acc[2] = fmin(acc[2], l ? xv[u] : INFINITY); acc[3] = fmax(acc[3], l ? xv[u] : -INFINITY);
      acc[4] = fmin(acc[4], l ? yv[u] : INFINITY); acc[5] = fmax(acc[5], l ? yv[u] : -INFINITY);
      acc[6] += x; acc[7] += y;
      acc[8] += ww; acc[9] = fma(ww, x, acc[9]); acc[10] = fma(ww, y, acc[10]); acc[11] = fma(ww, o, acc[11]);
      acc[12] = fmin(acc[12], l ? ov[u] : INFINITY); acc[13] = fmax(acc[13], l ? ov[u] : -INFINITY);
    }
  }
  block_reduce_store<kRedSlots>(acc, ops, scratch + (int64_t)blockIdx.x * kRedSlots);
}

__global__ __launch_bounds__(kBlock) void k_stats_final(const double* scratch, const int nblocks, double* out) {
  const int ops[kRedSlots] = {RSUM, RSUM, RMIN, RMAX, RMIN, RMAX, RSUM, RSUM,
                              RSUM, RSUM, RSUM, RSUM, RMIN, RMAX, RSUM, RSUM};
  fold_slot(scratch, nblocks, kRedSlots, ops[blockIdx.x], out);
}

constexpr int kSumSlots = 8;

// detector read-out fused with its reductions: one pass over the bundle, statistics accumulated in registers
__global__ __launch_bounds__(kBlock) void k_detector_readout(const ArtDetectorDesc d, const ArtBundleView b,
                                                             const double* w, const int64_t n, const double cx,
                                                             const double cy, const double co, double* p3x, double* p3y,
                                                             double* p3z, double* X, double* Y, double* opl,
                                                             double* scratch) {
  const int ops[kReadoutSlots] = {RSUM, RSUM, RMIN, RMAX, RMIN, RMAX, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM,
                                  RMIN, RMAX, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM};
  double acc[kReadoutSlots];
#pragma unroll
  for (int k = 0; k < kReadoutSlots; ++k) acc[k] = (ops[k] == RSUM) ? 0.0 : (ops[k] == RMIN ? INFINITY : -INFINITY);
  // Two consecutive slots per lane and iteration: every stream is read with one 16-byte access per lane (1 KiB per
  // wave instruction) and the three outputs are written the same way; per-dword range checking of the raw buffer
  // descriptors takes care of an odd tail.  Dead rays contribute nothing through selects (no divergent skip); their
  // output slots receive unspecified values.
  const BundleRsrc bi = make_rsrc(b, n);
  const unsigned nb8 = (unsigned)(n * 8);
  const __amdgpu_buffer_rsrc_t rw = rsrc_of(const_cast<double*>(w), w ? nb8 : 0u);
  const __amdgpu_buffer_rsrc_t r3x = rsrc_of(p3x, p3x ? nb8 : 0u), r3y = rsrc_of(p3y, p3y ? nb8 : 0u),
                               r3z = rsrc_of(p3z, p3z ? nb8 : 0u), rX = rsrc_of(X, X ? nb8 : 0u),
                               rY = rsrc_of(Y, Y ? nb8 : 0u), rO = rsrc_of(opl, opl ? nb8 : 0u);
  const int64_t npairs = (n + 1) / 2;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < npairs; j += stride) {
    const unsigned o16 = (unsigned)j * 16u, o2 = (unsigned)j * 2u;
    const D2 ox = ld_2f64(bi.ox, o16), oy = ld_2f64(bi.oy, o16), oz = ld_2f64(bi.oz, o16);
    const D2 dx = ld_2f64(bi.dx, o16), dy = ld_2f64(bi.dy, o16), dz = ld_2f64(bi.dz, o16);
    const D2 pa = ld_2f64(bi.path, o16), wv = ld_2f64(rw, o16);
    // two byte loads: a 16-bit load straddling the end of an odd-length array is dropped as a whole
    const unsigned char al[2] = {__builtin_amdgcn_raw_buffer_load_b8(bi.alive, (int)o2, 0, ART_LD_AUX),
                                 __builtin_amdgcn_raw_buffer_load_b8(bi.alive, (int)o2 + 1, 0, ART_LD_AUX)};
    double xo[2], yo[2], oo[2], p3[3][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      art::Ray r;
      r.ox = h ? ox.b : ox.a; r.oy = h ? oy.b : oy.a; r.oz = h ? oz.b : oz.a;
      r.dx = h ? dx.b : dx.a; r.dy = h ? dy.b : dy.a; r.dz = h ? dz.b : dz.a;
      r.path = h ? pa.b : pa.a;
      const bool live = al[h] != 0;
      double x, y, o;
      art::detector_ray(d, r, p3[0][h], p3[1][h], p3[2][h], x, y, o);
      xo[h] = x; yo[h] = y; oo[h] = o;
      readout_accumulate(acc, live, x, y, o, h ? wv.b : wv.a, w != nullptr, cx, cy, co);
    }
    st_2f64(r3x, o16, p3[0][0], p3[0][1]); st_2f64(r3y, o16, p3[1][0], p3[1][1]); st_2f64(r3z, o16, p3[2][0], p3[2][1]);
    st_2f64(rX, o16, xo[0], xo[1]); st_2f64(rY, o16, yo[0], yo[1]); st_2f64(rO, o16, oo[0], oo[1]);
  }
#ifdef ART_READOUT_SHUFFLE_REDUCE      // round-1 form: 24 shuffle trees per wave + __syncthreads()
  block_reduce_store<kReadoutSlots>(acc, ops, scratch + (int64_t)blockIdx.x * kReadoutSlots);
#else
  // One partial per WORKGROUP, [block][24] as k_readout_final expects it.  Per wave the LDS transpose of the fused tail
  // (wave_reduce24, ~100 instructions instead of 24 shuffle trees), then the four waves meet in LDS behind a bare
  // s_barrier: __syncthreads() would also wait for the acknowledgement of every store this workgroup has in flight
  // (vmcnt).  174 -> 158 us per 1e7 rays (tools/r02_exp21.sh).
  __shared__ double s_tile[(kBlock / 64) * 8 * kTileStride];
  __shared__ double s_part[kBlock / 64][kReadoutSlots];
  const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
  double tot[3];
  wave_reduce24(acc, s_tile + wv_ * (8 * kTileStride), lane, tot);
  if ((lane & 7) == 0) {
    const int stat = lane >> 3;
    int g0 = 0, g1 = 0, g2 = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      g0 = (stat == q) ? kPassSlot[0][q] : g0;
      g1 = (stat == q) ? kPassSlot[1][q] : g1;
      g2 = (stat == q) ? kPassSlot[2][q] : g2;
    }
    s_part[wv_][g0] = tot[0];
    s_part[wv_][g1] = tot[1];
    if (stat < 6) s_part[wv_][g2] = tot[2];
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (threadIdx.x < kUsedSlots) {
    const int k = threadIdx.x;
    double v = s_part[0][k];
#pragma unroll
    for (int j = 1; j < kBlock / 64; ++j)
      v = (ops[k] == RSUM) ? v + s_part[j][k] : (ops[k] == RMIN ? fmin(v, s_part[j][k]) : fmax(v, s_part[j][k]));
    scratch[(int64_t)blockIdx.x * kReadoutSlots + k] = v;
  } else if (threadIdx.x < kReadoutSlots) {
    scratch[(int64_t)blockIdx.x * kReadoutSlots + threadIdx.x] = 0.0;    // reserved slots 22, 23
  }
#endif
}

constexpr int kScanSlots = 33;

// the slot's seven streams + weight of kU grid-stride iterations, requested up front (see ld_nt)
template <int kU>
struct SlotBatch {
  art::Ray r[kU];
  double w[kU];
  bool live[kU];
};
// every stream of the batch, unconditionally (a lane beyond the end re-reads slot n - 1 and is dead)
template <int kU, bool HAS_W>
__device__ __forceinline__ void load_batch_all(const ArtBundleView& b, const double* w, const int64_t i0, const int64_t stride,
                                               const int64_t n, SlotBatch<kU>& q) {
  const int64_t last = n - 1;
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int64_t i = i0 + u * stride, c = i < n ? i : last;
    q.live[u] = (ld_nt(b.alive + c) != 0) & (i < n);
    q.r[u].ox = ld_nt(b.ox + c); q.r[u].oy = ld_nt(b.oy + c); q.r[u].oz = ld_nt(b.oz + c);
    q.r[u].dx = ld_nt(b.dx + c); q.r[u].dy = ld_nt(b.dy + c); q.r[u].dz = ld_nt(b.dz + c);
    q.r[u].path = ld_nt(b.path + c);
    q.w[u] = HAS_W ? ld_nt(w + c) : 1.0;
  }
}
template <bool HAS_W>
__global__ __launch_bounds__(kBlock) void k_scan_moments_partial(const ArtDetectorDesc d, const ArtBundleView b,
                                                                 const double* w, const int64_t n, const double co,
                                                                 const double span, double* scratch) {
  double acc[kScanSlots];
#pragma unroll
  for (int k = 0; k < kScanSlots; ++k) acc[k] = 0.0;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  constexpr int kU = 1;     // (two slots per iteration need 148 VGPRs = 3 waves per SIMD: 0.65 of peak; one: 4 waves)
  for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < n; i0 += kU * stride) {
    SlotBatch<kU> q;
    load_batch_all<kU, HAS_W>(b, w, i0, stride, n, q);
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const bool l = q.live[u];
      double q0[3], sq[3];
      bool crosses;
      art::detector_ray_scan(d, q.r[u], span, q0[0], q0[1], q0[2], sq[0], sq[1], sq[2], crosses);
      acc[32] += (l && crosses) ? 1.0 : 0.0;
      q0[2] -= co;
      sq[2] -= 1.0;
      const double ww = l ? q.w[u] : 0.0;
      acc[0] += l ? 1.0 : 0.0;
      acc[16] += ww;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int o = 1 + 5 * k;
        const double a0 = l ? q0[k] : 0.0, s0 = l ? sq[k] : 0.0;       // (a dead slot's values are unspecified: selected away)
        acc[o] += a0; acc[o + 1] += s0;
        acc[o + 2] = fma(a0, a0, acc[o + 2]); acc[o + 3] = fma(a0, s0, acc[o + 3]);
        acc[o + 4] = fma(s0, s0, acc[o + 4]);
        const double wq = ww * a0, ws = ww * s0;
        acc[16 + o] += wq; acc[16 + o + 1] += ws;
        acc[16 + o + 2] = fma(wq, a0, acc[16 + o + 2]); acc[16 + o + 3] = fma(wq, s0, acc[16 + o + 3]);
        acc[16 + o + 4] = fma(ws, s0, acc[16 + o + 4]);
      }
    }
  }
  block_reduce_store_f<kScanSlots>(acc, [](int) { return (int)RSUM; }, scratch + (int64_t)blockIdx.x * kScanSlots);
}

__global__ __launch_bounds__(kBlock) void k_scan_moments_final(const double* scratch, const int nblocks, double* out) {
  fold_slot(scratch, nblocks, kScanSlots, RSUM, out);
}

__global__ __launch_bounds__(kBlock) void k_readout_final(const double* scratch, const int nblocks, double* out) {
  const int ops[kReadoutSlots] = {RSUM, RSUM, RMIN, RMAX, RMIN, RMAX, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM,
                                  RMIN, RMAX, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM};
  fold_slot(scratch, nblocks, kReadoutSlots, ops[blockIdx.x], out);
}

template <bool HAS_W>
__global__ __launch_bounds__(kBlock) void k_moments_partial(const uint8_t* alive, const double* X, const double* Y,
                                                            const double* opl, const double* w, const int64_t n,
                                                            const double cx, const double cy, const double co,
                                                            double* scratch) {
  const int ops[kSumSlots] = {RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM};
  double acc[kSumSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * kBlock, last = n - 1;
  for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < n; i0 += kRedUnroll * stride) {
    double xv[kRedUnroll], yv[kRedUnroll], ov[kRedUnroll], wv[kRedUnroll];
    bool live[kRedUnroll];
#pragma unroll
    for (int u = 0; u < kRedUnroll; ++u) {
      const int64_t i = i0 + u * stride, c = i < n ? i : last;
      live[u] = (ld_nt(alive + c) != 0) & (i < n);
      xv[u] = X ? ld_nt(X + c) : cx;       // (wave-uniform: an absent stream sits on its centre)
      yv[u] = Y ? ld_nt(Y + c) : cy;
      ov[u] = opl ? ld_nt(opl + c) : co;
      wv[u] = HAS_W ? ld_nt(w + c) : 1.0;
    }
#pragma unroll
    for (int u = 0; u < kRedUnroll; ++u) {
      const bool l = live[u];
      const double ww = l ? wv[u] : 0.0;
      const double ex = (l && X) ? xv[u] - cx : 0.0, ey = (l && Y) ? yv[u] - cy : 0.0, eo = (l && opl) ? ov[u] - co : 0.0;
      acc[0] += ww;
      acc[1] = fma(ww * ex, ex, acc[1]);
      acc[2] = fma(ww * ey, ey, acc[2]);
      acc[3] = fma(ww * eo, eo, acc[3]);
      acc[4] += l ? 1.0 : 0.0;
    }
  }
  block_reduce_store<kSumSlots>(acc, ops, scratch + (int64_t)blockIdx.x * kSumSlots);
}

template <bool HAS_W>
__global__ __launch_bounds__(kBlock) void k_bundle_sums_partial(const ArtBundleView b, const double* w,
                                                                const int64_t n, double* scratch) {
  const int ops[kSumSlots] = {RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM};
  double acc[kSumSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * kBlock, last = n - 1;
  for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < n; i0 += kRedUnroll * stride) {
    double v[kRedUnroll][7];
    bool live[kRedUnroll];
#pragma unroll
    for (int u = 0; u < kRedUnroll; ++u) {
      const int64_t i = i0 + u * stride, c = i < n ? i : last;
      live[u] = (ld_nt(b.alive + c) != 0) & (i < n);
      v[u][0] = ld_nt(b.ox + c); v[u][1] = ld_nt(b.oy + c); v[u][2] = ld_nt(b.oz + c);
      v[u][3] = ld_nt(b.dx + c); v[u][4] = ld_nt(b.dy + c); v[u][5] = ld_nt(b.dz + c);
      v[u][6] = HAS_W ? ld_nt(w + c) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kRedUnroll; ++u) {
      acc[0] += live[u] ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < 7; ++k) acc[1 + k] += live[u] ? v[u][k] : 0.0;
    }
  }
  block_reduce_store<kSumSlots>(acc, ops, scratch + (int64_t)blockIdx.x * kSumSlots);
}

__global__ __launch_bounds__(kBlock) void k_sums_final(const double* scratch, const int nblocks, double* out) {
  fold_slot(scratch, nblocks, kSumSlots, RSUM, out);
}

// ------------------------------------------------------------------------------------------- source weights
struct Axis3 { double x, y, z; };

__device__ __forceinline__ double angle_to_axis(const Axis3 ax, double vx, double vy, double vz) {
  // AngleBetweenTwoVectors(Axis, vector), ART/ModuleGeometry.py:40-44, with the norms as the reference applies them
  const double u = sqrt(art::dot3(ax.x, ax.y, ax.z, ax.x, ax.y, ax.z)), v = sqrt(art::dot3(vx, vy, vz, vx, vy, vz));
  const double ax_ = ax.x * v - vx * u, ay_ = ax.y * v - vy * u, az_ = ax.z * v - vz * u;
  const double bx_ = ax.x * v + vx * u, by_ = ax.y * v + vy * u, bz_ = ax.z * v + vz * u;
  return 2.0 * atan2(sqrt(art::dot3(ax_, ay_, az_, ax_, ay_, az_)), sqrt(art::dot3(bx_, by_, bz_, bx_, by_, bz_)));
}

// tan^2(angle / 2) of the same Kahan pair, |a v - v' u|^2 / |a v + v' u|^2: monotone in the angle, one reciprocal instead of
// two square roots and an atan2 per ray -- for a MAXIMUM over rays the angle itself is formed once, from the largest ratio
// (k_analysis_fold: 2 atan(sqrt(.))).
__device__ __forceinline__ double tan2_half_angle_to_axis(const Axis3 ax, const double u, double vx, double vy, double vz,
                                                          const double v) {     // u = |axis|, v = |vector|
  const double ax_ = ax.x * v - vx * u, ay_ = ax.y * v - vy * u, az_ = ax.z * v - vz * u;
  const double bx_ = ax.x * v + vx * u, by_ = ax.y * v + vy * u, bz_ = ax.z * v + vz * u;
  // (the denominator is ~4 |u|^2 |v|^2 for the small angles this is about: a refined reciprocal, not an IEEE division)
  return art::dot3(ax_, ay_, az_, ax_, ay_, az_) * art::rcp_full(art::dot3(bx_, by_, bz_, bx_, by_, bz_));
}

// `axis_sums` (or NULL): DEVICE, the 8 sums of art_bundle_sums -- the axis is then the bundle's own central ray, mean vector
// normalised (FindCentralRay + the Ray.vector setter, ART/ModuleProcessing.py:464-482, ART/ModuleOpticalRay.py:85-90), formed
// here instead of on the host: no round trip between the sums and the weights
__device__ __forceinline__ Axis3 axis_of(const Axis3 given, const double* axis_sums) {
  if (axis_sums == nullptr) return given;
  const double c = axis_sums[0];
  const double x = axis_sums[4] / c, y = axis_sums[5] / c, z = axis_sums[6] / c;
  const double nrm = sqrt(art::dot3(x, y, z, x, y, z));
  return Axis3{x / nrm, y / nrm, z / nrm};
}

__global__ __launch_bounds__(kBlock) void k_gauss_max_partial(const ArtBundleView b, const Axis3 ax_in, const double* axis_sums,
                                                              const int64_t n, double* scratch) {
  const Axis3 ax = axis_of(ax_in, axis_sums);
  const int ops[kSumSlots] = {RMAX, RMAX, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM};
  double acc[kSumSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * kBlock, last = n - 1;
  constexpr int kU = 2;     // (an atan2 and three square roots per ray: the arithmetic hides the rest of the latency)
  for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < n; i0 += kU * stride) {
    double v[kU][6];
    bool live[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + u * stride, c = i < n ? i : last;
      live[u] = (ld_nt(b.alive + c) != 0) & (i < n);
      v[u][0] = ld_nt(b.ox + c); v[u][1] = ld_nt(b.oy + c); v[u][2] = ld_nt(b.oz + c);
      v[u][3] = ld_nt(b.dx + c); v[u][4] = ld_nt(b.dy + c); v[u][5] = ld_nt(b.dz + c);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      // a dead slot holds unspecified values: it takes part as the axis itself at the origin (angle 0, distance 0)
      const double dx = live[u] ? v[u][3] : ax.x, dy = live[u] ? v[u][4] : ax.y, dz = live[u] ? v[u][5] : ax.z;
      const double px = live[u] ? v[u][0] : 0.0, py = live[u] ? v[u][1] : 0.0, pz = live[u] ? v[u][2] : 0.0;
      const double ang = angle_to_axis(ax, dx, dy, dz);
      acc[0] = fmax(acc[0], live[u] ? ang : 0.0);
      acc[1] = fmax(acc[1], sqrt(art::dot3(px, py, pz, px, py, pz)));
    }
  }
  block_reduce_store<kSumSlots>(acc, ops, scratch + (int64_t)blockIdx.x * kSumSlots);
}

__global__ __launch_bounds__(kBlock) void k_gauss_max_final(const double* scratch, const int nblocks, double* out) {
  const int ops[kSumSlots] = {RMAX, RMAX, RSUM, RSUM, RSUM, RSUM, RSUM, RSUM};
  double acc[kSumSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int blk = threadIdx.x; blk < nblocks; blk += kBlock) {
    acc[0] = fmax(acc[0], scratch[(int64_t)blk * kSumSlots + 0]);
    acc[1] = fmax(acc[1], scratch[(int64_t)blk * kSumSlots + 1]);
  }
  block_reduce_store<kSumSlots>(acc, ops, out);
}

__global__ __launch_bounds__(kBlock) void k_gauss_weights(const ArtBundleView b, const Axis3 ax_in, const double* axis_sums,
                                                          const double kexp, const double* maxima, const int64_t n, double* w) {
  const Axis3 ax = axis_of(ax_in, axis_sums);
  const double div = maxima[0], maxdist = maxima[1];
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    double q;
    if (div > 1e-12) {   // diverging bundle: profile in angle (ModuleSource.py:247-251)
      q = tan(angle_to_axis(ax, b.dx[i], b.dy[i], b.dz[i])) / div;
    } else {             // collimated: profile in distance from the origin (:252-259)
      q = sqrt(art::dot3(b.ox[i], b.oy[i], b.oz[i], b.ox[i], b.oy[i], b.oz[i])) / maxdist;
    }
    w[i] = exp(-2.0 * q * q * kexp);
  }
}

// ------------------------------------------------------------------------------------------- guide rays
// One guide ray per thread, each through ITS OWN element (art_trace_guides): the 1-ray alignment traces of OEPlacement for
// up to 8 chains in one launch.  The descriptors are kernel arguments, indexed per lane (vector loads from the constant
// argument segment: this kernel is launch-latency, not throughput).
struct GuideArgs {
  ArtElementDesc e[ART_GUIDES_MAX];
};
template <bool DEFECT>
__global__ __launch_bounds__(64) void k_trace_guides(const GuideArgs ga, double* rays, uint8_t* alive, const int count) {
  const int t = threadIdx.x;
  if (t >= count) return;
  art::Ray r;
  double* q = rays + 8 * t;
  r.ox = q[0]; r.oy = q[1]; r.oz = q[2]; r.dx = q[3]; r.dy = q[4]; r.dz = q[5]; r.path = q[6]; r.inc = q[7];
  bool ok = alive[t] != 0;
  if (ok) {
    // one lane at a time: the torus solver's Newton loop votes across the wave and the descriptor must be wave-uniform
    for (int j = 0; j < count; ++j) {
      if (t == j) ok = art::trace_ray_dyn<DEFECT>(ga.e[j], ga.e[j].zern, r);
    }
  }
  if (ok) { q[0] = r.ox; q[1] = r.oy; q[2] = r.oz; q[3] = r.dx; q[4] = r.dy; q[5] = r.dz; q[6] = r.path; q[7] = r.inc; }
  alive[t] = (uint8_t)(ok ? 1 : 0);
}

// ------------------------------------------------------------------------------------------- batched analysis
// art_analyse_bundles: blockIdx.y = job.  Partials per job and workgroup, folded in a fixed order.
constexpr int kAnaMom = 42;      // 32 moment sums, kink shifts (2), max angle, bounding box + path range (6), pad
constexpr int kAnaPlace = 24;    // per job: ArtDetectorDesc (15 doubles), axis (3), co, pad
constexpr int kAnaBlocks = 1024; // workgroups per job of the moments pass at most
constexpr int kAnaSumsPad = 16;  // doubles per job of the folded sums (9 used)
// Workgroups per job of the moments pass: a function of the ray count ALONE, so that a bundle's moments are folded in the
// same order whether it is analysed alone or as one of a list (same bits either way).
inline int analysis_blocks(int64_t n) {
  const int64_t b = (n + kBlock - 1) / kBlock;
  return (int)(b < 1 ? 1 : (b > kAnaBlocks ? kAnaBlocks : b));
}
inline int64_t analysis_tiles(int64_t n) { return n < 1 ? 1 : (n + kBlock - 1) / kBlock; }
// ART_ANALYSIS_ORDER=job: the moments pass in job-major order instead of XCD-grouped (read per call: in-process A/B)
inline int analysis_job_major() {
  const char* e = getenv("ART_ANALYSIS_ORDER");
  return (e && e[0] == 'j') ? 1 : 0;
}
__device__ __forceinline__ int ana_mom_op(const int q) {   // operator of moment-pass partial q
  return (q < 32) ? RSUM : ((q == 33 || q == 35 || q == 37 || q == 39) ? RMIN : ((q == 41) ? RSUM : RMAX));
}
__device__ __forceinline__ int ana_mom_slot(const int q) { // where partial q lands in a job's output row
  return (q < 32) ? 20 + q : ((q < 35) ? 53 + (q - 32) : 56 + (q - 35));
}

// Pass (1) for the jobs that do not bring their sums along (ArtAnalysisJob.sums): one workgroup per tile of 256 slots, one
// ray per lane, the canonical order of run_sums8.  rows: [job][kSumRows][ntiles] (+ the chunk totals of a two-stage fold).
// (launched per RUN of consecutive jobs that need it -- job = job0 + blockIdx.y: a grid over all jobs would dispatch 39 063
// workgroups per 1e7-ray job only to let them return)
__global__ __launch_bounds__(kBlock) void k_analysis_sums(const ArtAnalysisJob* __restrict__ jobs, const int job0,
                                                          const int64_t n, double* rows, const int64_t job_stride) {
  const int job = job0 + (int)blockIdx.y;
  const ArtAnalysisJob& jb = jobs[job];
  if (jb.sums != nullptr) return;      // (workgroup-uniform) formed by the tracing launch already
  __shared__ __attribute__((aligned(16))) double s_tile[(kBlock / 64) * 8 * kTileStride];
  __shared__ double s_run[(kBlock / 64) * kSumRows];
  const unsigned t = threadIdx.x;
  // buffer descriptors (n <= 2^28, checked by the host): slots beyond the end read alive = 0, and a dead slot's data is
  // requested at an out-of-range offset, i.e. not at all (a workgroup of dead slots -- the shadow of a mask -- moves 256 bytes)
  const unsigned i = blockIdx.x * kBlock + t;
  const BundleRsrc rb = make_rsrc(jb.b, n);
  const bool live = __builtin_amdgcn_raw_buffer_load_b8(rb.alive, (int)i, 0, ART_LD_AUX) != 0;
  const unsigned o8 = live ? i * 8u : kDropOffset;
  art::Ray r;
  r.ox = ld_f64(rb.ox, o8); r.oy = ld_f64(rb.oy, o8); r.oz = ld_f64(rb.oz, o8);
  r.dx = ld_f64(rb.dx, o8); r.dy = ld_f64(rb.dy, o8); r.dz = ld_f64(rb.dz, o8);
  r.path = ld_f64(rb.path, o8);
  const double w = jb.w ? ld_f64(rsrc_of(const_cast<double*>(jb.w), (unsigned)(n * 8)), o8) : 1.0;
  double v[8];
  sums_values(v, live, r, w);
  const double tot = run_sums8(v, s_tile + (t >> 6) * (8 * kTileStride), t & 63);
  tile_sums_store(tot, __ballot(live), s_run, t, rows + (int64_t)job * job_stride, gridDim.x, blockIdx.x);
}
// grids (kSumRows, kFoldChunks, jobs) and (kSumRows, 1, jobs): the folds of launch_sums_fold_*, or a copy of the sums a job
// brought along
__global__ __launch_bounds__(kFoldBlock) void k_analysis_sums_fold1(const ArtAnalysisJob* __restrict__ jobs, const int job0,
                                                                    double* rows, const int64_t job_stride,
                                                                    const int64_t ntiles) {
  const int job = job0 + (int)blockIdx.z;
  if (jobs[job].sums != nullptr) return;
  sums_fold1(rows + (int64_t)job * job_stride, ntiles);
}
__global__ __launch_bounds__(kFoldBlock) void k_analysis_sums_fold2(const ArtAnalysisJob* __restrict__ jobs, double* rows,
                                                                    const int64_t job_stride, const int64_t ntiles,
                                                                    const int direct, double* sums) {
  double* out = sums + (int64_t)blockIdx.z * kAnaSumsPad;
  const double* given = jobs[blockIdx.z].sums;
  if (given != nullptr) {
    if (threadIdx.x == 0) out[blockIdx.x] = given[blockIdx.x];
    return;
  }
  sums_fold2(rows + (int64_t)blockIdx.z * job_stride, ntiles, out, direct);
}

// one thread per job: the sums are folded; place the detector
__global__ __launch_bounds__(64) void k_analysis_place(const ArtAnalysisJob* __restrict__ jobs, const int n_jobs,
                                                       const double* sums, double* place, double* out) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= n_jobs) return;
  const double* s_sum = sums + (int64_t)j * kAnaSumsPad;
  const ArtAnalysisJob& jb = jobs[j];
  double* o = out + (int64_t)j * ART_ANALYSIS_DOUBLES;
  double* pl = place + (int64_t)j * kAnaPlace;
  for (int k = 0; k < ART_ANALYSIS_DOUBLES; ++k) o[k] = 0.0;
  for (int k = 0; k < kSumRows; ++k) o[k] = s_sum[k];
  o[53] = -INFINITY; o[54] = INFINITY;
  o[56] = INFINITY; o[57] = -INFINITY; o[58] = INFINITY; o[59] = -INFINITY; o[60] = INFINITY; o[61] = -INFINITY;
  if (jb.mode == ART_JOB_SUMS) return;
  ArtDetectorDesc d;
  double ref[3], axis[3], co;
  if (s_sum[0] > 0.0) {
    art::analysis_place(s_sum, jb.mode, jb.distance, jb.centre, jb.normal, jb.refpoint, d, ref, axis, co);
  } else {
    for (int k = 0; k < 3; ++k) { d.centre[k] = NAN; d.normal[k] = NAN; ref[k] = NAN; axis[k] = NAN; }
    for (int k = 0; k < 9; ++k) d.rot[k] = NAN;
    co = NAN;
  }
  for (int k = 0; k < 3; ++k) { o[10 + k] = d.centre[k]; o[13 + k] = d.normal[k]; o[16 + k] = ref[k]; }
  o[19] = co;
  for (int k = 0; k < 3; ++k) { pl[k] = d.centre[k]; pl[3 + k] = d.normal[k]; pl[15 + k] = axis[k]; }
  for (int k = 0; k < 9; ++k) pl[6 + k] = d.rot[k];
  pl[18] = co;
}

// The moments pass keeps 42 accumulators (84 VGPRs) beside the ray: 128 VGPRs = 4 waves per SIMD, which is also all the
// fixed grid of 1024 workgroups offers.  What it gained in round 5: the next slot's nine streams are requested
// UNCONDITIONALLY (ld_nt, selects instead of the alive-guarded skip) while the current slot is worked on.
template <bool HAS_W>
__device__ __forceinline__ void moments_body(const ArtAnalysisJob& jb, const int64_t n, const double* pl, double (&acc)[kAnaMom],
                                             const unsigned blk, const unsigned nblk) {
  ArtDetectorDesc d;
#pragma unroll
  for (int k = 0; k < 3; ++k) { d.centre[k] = pl[k]; d.normal[k] = pl[3 + k]; }
#pragma unroll
  for (int k = 0; k < 9; ++k) d.rot[k] = pl[6 + k];
  const Axis3 ax = {pl[15], pl[16], pl[17]};
  const double au = sqrt(art::dot3(ax.x, ax.y, ax.z, ax.x, ax.y, ax.z));      // |axis|: once, not per ray
  const double inv_nn = 1.0 / art::dot3(d.normal[0], d.normal[1], d.normal[2], d.normal[0], d.normal[1], d.normal[2]);
  const double co = pl[18];
  const ArtBundleView b = jb.b;
  const double* w = jb.w;
  const int64_t stride = (int64_t)nblk * kBlock;
  // Every stream of the slot is requested UNCONDITIONALLY at the top of its iteration (alive is applied by selects: no
  // load waits for another load); buffer descriptors: one 32-bit offset register serves all nine streams (nine 64-bit
  // addresses would not fit beside the accumulators), slots beyond the end read 0 = dead.  A software pipeline (the next
  // slot's loads in flight behind this slot's arithmetic) does NOT fit: 41 fp64 accumulators + two slots' state exceed
  // the 128 registers of 4 waves per SIMD and the compiler parks the prefetched values in scratch (tried, round 5).
  // n <= 2^28 (checked by the host).
  const BundleRsrc rb = make_rsrc(b, n);
  const __amdgpu_buffer_rsrc_t rw = rsrc_of(const_cast<double*>(w), HAS_W ? (unsigned)(n * 8) : 0u);
  const unsigned ustride = (unsigned)stride, un_ = (unsigned)n;
  int cnt = 0;
  unsigned i = blk * kBlock + threadIdx.x;
  bool l = __builtin_amdgcn_raw_buffer_load_b8(rb.alive, (int)i, 0, ART_LD_AUX) != 0;
  for (; i < un_; i += ustride) {
    // the data of an ALIVE slot only (a dead one is requested out of range: no traffic; its fields read 0 and are selected
    // away below), and the NEXT slot's alive byte in flight behind it
    const unsigned o8 = l ? i * 8u : kDropOffset;
    const unsigned char nl_raw = __builtin_amdgcn_raw_buffer_load_b8(rb.alive, (int)(i + ustride), 0, ART_LD_AUX);
    const double cox = ld_f64(rb.ox, o8), coy = ld_f64(rb.oy, o8), coz = ld_f64(rb.oz, o8);
    const double cdx = ld_f64(rb.dx, o8), cdy = ld_f64(rb.dy, o8), cdz = ld_f64(rb.dz, o8);
    const double pth = ld_f64(rb.path, o8);
    const double wv = HAS_W ? ld_f64(rw, o8) : 1.0;
    art::Ray r;
    r.ox = cox; r.oy = coy; r.oz = coz; r.dx = cdx; r.dy = cdy; r.dz = cdz; r.path = pth; r.inc = 0.0;
    double q0[3], sq[3], sk, un;
    art::detector_ray_scan_kink(d, r, inv_nn, q0[0], q0[1], q0[2], sq[0], sq[1], sq[2], sk, un);
    acc[35] = fmin(acc[35], l ? q0[0] : INFINITY); acc[36] = fmax(acc[36], l ? q0[0] : -INFINITY);
    acc[37] = fmin(acc[37], l ? q0[1] : INFINITY); acc[38] = fmax(acc[38], l ? q0[1] : -INFINITY);
    acc[39] = fmin(acc[39], l ? q0[2] : INFINITY); acc[40] = fmax(acc[40], l ? q0[2] : -INFINITY);
    acc[32] = fmax(acc[32], (l && sk <= 0.0) ? sk : -INFINITY);
    acc[33] = fmin(acc[33], (l && sk > 0.0) ? sk : INFINITY);
    const double t2 = tan2_half_angle_to_axis(ax, au, r.dx, r.dy, r.dz, un);
    acc[34] = fmax(acc[34], l ? t2 : 0.0);
    q0[2] -= co;
    sq[2] -= 1.0;
    const double ww = l ? wv : 0.0;
    cnt += l ? 1 : 0;
    acc[16] += ww;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int o = 1 + 5 * k;
      const double a0 = l ? q0[k] : 0.0, s0 = l ? sq[k] : 0.0;       // (a dead slot's values are unspecified: selected away)
      acc[o] += a0; acc[o + 1] += s0;
      acc[o + 2] = fma(a0, a0, acc[o + 2]); acc[o + 3] = fma(a0, s0, acc[o + 3]);
      acc[o + 4] = fma(s0, s0, acc[o + 4]);
      const double wq = ww * a0, ws = ww * s0;
      acc[16 + o] += wq; acc[16 + o + 1] += ws;
      acc[16 + o + 2] = fma(wq, a0, acc[16 + o + 2]); acc[16 + o + 3] = fma(wq, s0, acc[16 + o + 3]);
      acc[16 + o + 4] = fma(ws, s0, acc[16 + o + 4]);
    }
    l = nl_raw != 0;
  }
  acc[0] = (double)cnt;      // (a lane folds < 2^31 slots: the integer count is exact, as the sum of 1.0s was)
}

// 1-D grid of (P rounded up to a multiple of 8) x J workgroups, XCD-grouped like the scene launches (scene_wg): id = 8 k + xcd,
// job = k mod J, block = 8 (k / J) + xcd -- the J workgroups that walk the SAME slots of J bundles run on one XCD, so the
// weights the jobs of a loop list share (one source: one intensity array) are fetched into its L2 once, not J times.
// Which slots a block folds, and the fold order, do not depend on the mapping.
__global__ __launch_bounds__(kBlock, 4) void k_analysis_moments(const ArtAnalysisJob* __restrict__ jobs, const int64_t n,
                                                             const double* place, const double* out, double* scratch,
                                                             const int P, const int job0, const int J, const int job_major) {
  const unsigned id = blockIdx.x, P8 = gridDim.x / (unsigned)J;     // (jobs job0 .. job0 + J - 1 in this launch)
  unsigned blk;
  int j;
  if (job_major) {                           // (ART_ANALYSIS_ORDER=job: one job's workgroups after the other, for the A/B)
    j = (int)(id / P8);
    blk = id - (unsigned)j * P8;
  } else {
    const unsigned k = id >> 3, kq = k / (unsigned)J;
    blk = kq * 8u + (id & 7u);
    j = (int)(k - kq * (unsigned)J);
  }
  if (blk >= (unsigned)P) return;            // (padding)
  j += job0;
  const ArtAnalysisJob& jb = jobs[j];
  double acc[kAnaMom];
#pragma unroll
  for (int k = 0; k < kAnaMom; ++k) acc[k] = (ana_mom_op(k) == RSUM) ? 0.0 : (ana_mom_op(k) == RMIN ? INFINITY : -INFINITY);
  acc[34] = 0.0;    // the largest angle of an empty set is 0 (ReturnNumericalAperture's max over nothing never happens)
  const bool active = jb.mode != ART_JOB_SUMS && out[(int64_t)j * ART_ANALYSIS_DOUBLES] > 0.0;
  if (active) {
    if (jb.w) moments_body<true>(jb, n, place + (int64_t)j * kAnaPlace, acc, blk, (unsigned)P);
    else moments_body<false>(jb, n, place + (int64_t)j * kAnaPlace, acc, blk, (unsigned)P);
  }
  block_reduce_store_f<kAnaMom>(acc, [](int q) { return ana_mom_op(q); }, scratch + ((int64_t)j * P + blk) * kAnaMom);
}

// grid (kAnaMom - 1, n_jobs): workgroup (q, j) folds partial q of job j into its output slot
__global__ __launch_bounds__(kBlock) void k_analysis_fold(const ArtAnalysisJob* __restrict__ jobs, const int nblocks,
                                                          const double* scratch, double* out) {
  const int q = blockIdx.x, j = blockIdx.y;
  if (jobs[j].mode == ART_JOB_SUMS) return;
  const int op = ana_mom_op(q);
  const int op1[1] = {op};
  const double ident = (op == RSUM) ? 0.0 : (op == RMIN ? INFINITY : -INFINITY);
  double acc[1] = {ident};
  const double* mine = scratch + (int64_t)j * nblocks * kAnaMom;
  constexpr int kU = 4;    // partials in flight per thread (the fold order stays blk, blk + 256, ...)
  for (int blk = threadIdx.x; blk < nblocks; blk += kBlock * kU) {
    double v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) v[u] = (blk + u * kBlock < nblocks) ? mine[(int64_t)(blk + u * kBlock) * kAnaMom + q] : ident;
#pragma unroll
    for (int u = 0; u < kU; ++u) acc[0] = (op == RSUM) ? acc[0] + v[u] : (op == RMIN ? fmin(acc[0], v[u]) : fmax(acc[0], v[u]));
  }
  __shared__ double s_res[1];
  block_reduce_store<1>(acc, op1, s_res);
  __syncthreads();
  // partial 34 carries tan^2(angle / 2) of the ray farthest from the mean direction: the angle is formed here, once
  if (threadIdx.x == 0) out[(int64_t)j * ART_ANALYSIS_DOUBLES + ana_mom_slot(q)] = (q == 34) ? 2.0 * atan(sqrt(s_res[0])) : s_res[0];
}

// ------------------------------------------------------------------------------------------- compaction
// Stable stream compaction of the alive mask in three passes: per-tile counts, exclusive scan of the tile
// counts by one workgroup, scatter with an intra-wave ballot/popcount rank.
constexpr int kTile = 2048;  // slots per workgroup: 8 per lane

// number of non-zero bytes of a 64-bit word
__device__ __forceinline__ int nonzero_bytes(const unsigned long long x) {
  const unsigned long long lo7 = 0x7f7f7f7f7f7f7f7full;
  return __popcll(((x & lo7) + lo7 | x) & ~lo7);
}
// A tile's 2048 alive bytes as ONE 8-byte load per lane (the count needs no order); the last, partial tile and a mask that
// does not start on an 8-byte boundary take the byte path.  10 MB per 1e7 rays: the kernel is a few microseconds of launch
// and latency, not bandwidth (it was 14.6 us with eight dependent byte loads per lane).
__global__ __launch_bounds__(kBlock) void k_compact_count(const uint8_t* alive, const int64_t n, int32_t* counts) {
  __shared__ int s[kBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kTile;
  int c = 0;
  if (base + kTile <= n && (reinterpret_cast<uintptr_t>(alive) & 7u) == 0) {     // (workgroup-uniform)
    c = nonzero_bytes(reinterpret_cast<const unsigned long long*>(alive + base)[threadIdx.x]);   // (default policy: the scatter pass reads the mask again)
  } else {
    uint8_t a[kTile / kBlock];
#pragma unroll
    for (int j = 0; j < kTile / kBlock; ++j) {
      const int64_t i = base + j * kBlock + threadIdx.x;
      a[j] = (i < n) ? alive[i] : 0;
    }
#pragma unroll
    for (int j = 0; j < kTile / kBlock; ++j) c += a[j] != 0 ? 1 : 0;
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(1024) void k_compact_scan(int32_t* counts, const int64_t ntiles, int64_t* total,
                                                       int64_t* tile_offsets) {
  // single workgroup; each lane owns a contiguous chunk -> local sums -> LDS scan -> write back
  __shared__ int64_t s[1024];
  const int64_t per = (ntiles + 1023) / 1024;
  const int64_t lo = (int64_t)threadIdx.x * per;
  const int64_t hi = (lo + per < ntiles) ? lo + per : ntiles;
  int64_t sum = 0;
  for (int64_t k = lo; k < hi; ++k) sum += counts[k];
  s[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int64_t v = (threadIdx.x >= (unsigned)off) ? s[threadIdx.x - off] : 0;
    __syncthreads();
    s[threadIdx.x] += v;
    __syncthreads();
  }
  int64_t run = s[threadIdx.x] - sum;  // exclusive prefix of this chunk
  for (int64_t k = lo; k < hi; ++k) {
    tile_offsets[k] = run;
    run += counts[k];
  }
  if (threadIdx.x == 1023) *total = s[1023];
}

// Ranks of a tile's alive slots, all eight passes at once: the eight alive bytes of a lane (slots base + j*256 + t) are
// requested together, every wave ballots its eight passes and parks the popcounts, ONE barrier, and each lane adds up what
// lies in front of it (passes before its own in full, earlier waves of its own pass, its rank inside the ballot).  Round 4
// did this pass by pass: eight dependent byte loads and sixteen barriers per tile.
struct TileRanks {
  bool a[kTile / kBlock];
  int pos[kTile / kBlock];     // position of slot j among the tile's alive slots (valid where a[j])
};
__device__ __forceinline__ void tile_ranks(const uint8_t* alive, const int64_t n, const int64_t base, int (*s_cnt)[kBlock / 64],
                                           TileRanks& q) {
  constexpr int P = kTile / kBlock;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint8_t raw[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int64_t i = base + j * kBlock + threadIdx.x;
    raw[j] = (i < n) ? alive[i] : 0;      // (default policy: the count pass has just read these lines)
  }
  int rank[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    q.a[j] = raw[j] != 0;
    const unsigned long long m = __ballot(q.a[j]);
    rank[j] = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[j][wv] = __popcll(m);
  }
  __syncthreads();
  int run = 0;
#pragma unroll
  for (int j = 0; j < P; ++j) {
    int before = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kBlock / 64; ++k) {
      const int c = s_cnt[j][k];
      before += (k < wv) ? c : 0;
      tot += c;
    }
    q.pos[j] = run + before + rank[j];
    run += tot;
  }
}

__global__ __launch_bounds__(kBlock) void k_compact_scatter(const uint8_t* alive, const int64_t n,
                                                            const int64_t* tile_offsets, int64_t* idx_out) {
  __shared__ int s_cnt[kTile / kBlock][kBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kTile;
  const int64_t run = tile_offsets[blockIdx.x];
  TileRanks q;
  tile_ranks(alive, n, base, s_cnt, q);
#pragma unroll
  for (int j = 0; j < kTile / kBlock; ++j)
    if (q.a[j]) idx_out[run + q.pos[j]] = base + j * kBlock + threadIdx.x;
}

// ------------------------------------------------------------------------------------------- survivor records
// Send buffer of the multi-GPU gather (art_pack_survivors, include/art_hip.h): a 16-byte header (count, flags) followed
// by the sections X[count], Y[count], path[count] (fp64) and number[count] (int32) of the SURVIVING rays in slot order.
// Pass 1 + 2 are the compaction's (per-tile counts, one-workgroup scan); the header is written from the scan's total,
// and pass 3 scatters the records instead of slot indices.
constexpr int64_t kSurvHeader = 16;
constexpr int64_t kSurvDense = 1;   // header flag: every slot alive and numbers implicit -> no number section

__global__ void k_survivor_header(const int64_t* total, const int64_t n, const int implicit_numbers, int64_t* header) {
  if (threadIdx.x == 0) {
    header[0] = *total;
    header[1] = (implicit_numbers && *total == n) ? kSurvDense : 0;
  }
}

__global__ __launch_bounds__(kBlock) void k_survivor_scatter(const uint8_t* alive, const int64_t n,
                                                             const int64_t* tile_offsets, const double* X,
                                                             const double* Y, const double* opl, const int64_t* number,
                                                             const int64_t first, const int64_t step,
                                                             unsigned char* send) {
  __shared__ int s_cnt[kTile / kBlock][kBlock / 64];
  const int64_t* header = reinterpret_cast<const int64_t*>(send);
  const int64_t count = header[0];
  const bool dense = (header[1] & kSurvDense) != 0;
  double* sx = reinterpret_cast<double*>(send + kSurvHeader);
  double* sy = sx + count;
  double* so = sy + count;
  int32_t* sn = reinterpret_cast<int32_t*>(so + count);
  const int64_t base = (int64_t)blockIdx.x * kTile;
  const int64_t run = tile_offsets[blockIdx.x];
  TileRanks q;
  tile_ranks(alive, n, base, s_cnt, q);
  // four passes at a time: their records are requested together (12-16 loads in flight per lane) and then stored -- with
  // the default cache policy: the records of a masked shard land at unaligned positions, and non-temporal partial-line
  // stores measured 8 % slower (98.5 -> 106.5 us per 1e7 dense slots, profiles/r05_experiments.md)
  constexpr int P = kTile / kBlock, H = 4;
#pragma unroll
  for (int j0 = 0; j0 < P; j0 += H) {
    double xv[H], yv[H], ov[H];
    int64_t nv[H];
#pragma unroll
    for (int u = 0; u < H; ++u) {
      const int64_t i = base + (j0 + u) * kBlock + threadIdx.x;
      if (q.a[j0 + u]) {
        xv[u] = ld_nt(X + i); yv[u] = ld_nt(Y + i); ov[u] = ld_nt(opl + i);
        nv[u] = (number && !dense) ? ld_nt(number + i) : first + i * step;
      }
    }
#pragma unroll
    for (int u = 0; u < H; ++u) {
      if (q.a[j0 + u]) {
        const int64_t p = run + q.pos[j0 + u];
        sx[p] = xv[u]; sy[p] = yv[u]; so[p] = ov[u];
        if (!dense) sn[p] = (int32_t)nv[u];
      }
    }
  }
}

// Zero-copy form of the send buffer (art_survivor_finish): the read-out wrote X, Y, path of ALL n slots straight into the
// dense layout's sections; if every slot is alive the buffer is complete once its header says so.  Otherwise the header
// says "unpacked" (flags bit 1): the sections hold slot-indexed values with holes, and the caller packs them elsewhere.
constexpr int64_t kSurvUnpacked = 2;
// xhdr (optional): the rank's contribution to the per-step header exchange of a sharded run -- [0], [1] = count, flags as
// int64 bit patterns, [2 .. 25] = the shard's 24 read-out statistics (zeros if none are given)
constexpr int kXhdrDoubles = 26;
__device__ __forceinline__ void xheader_store(const int64_t count, const int64_t flags, const double* stats24, double* xhdr) {
  const int t = threadIdx.x;
  if (t == 0) { reinterpret_cast<int64_t*>(xhdr)[0] = count; reinterpret_cast<int64_t*>(xhdr)[1] = flags; }
  if (t < kReadoutSlots) xhdr[2 + t] = stats24 ? stats24[t] : 0.0;
}
__global__ void k_survivor_finish(const double* stats24, const int64_t n, int64_t* header, double* xhdr) {
  const int64_t count = (int64_t)stats24[0];
  const int64_t flags = (count == n) ? kSurvDense : kSurvUnpacked;
  if (threadIdx.x == 0) { header[0] = count; header[1] = flags; }
  if (xhdr) xheader_store(count, flags, stats24, xhdr);
}
__global__ void k_survivor_xheader(const int64_t* header, const double* stats24, double* xhdr) {
  xheader_store(header[0], header[1], stats24, xhdr);
}

// ------------------------------------------------------------------------------------------- sources
__global__ __launch_bounds__(kBlock) void k_make_source(const int32_t kind, const double size, const ArtDetectorDesc rs,
                                                        const int64_t first, const int64_t step, const int64_t n,
                                                        const int64_t n_total, const ArtBundleView out) {
  // rs.rot = rotation ez -> axis, rs.centre = S (ArtDetectorDesc reused as a POD carrier); slot i = global ray
  // first + i * step (step 1: a contiguous shard; step = world size: a strided one)
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    art::Ray r;
    art::source_ray(kind, size, rs.rot, rs.centre, first + i * step, n_total, r);
    store_ray(out, i, r);
    out.alive[i] = 1;
  }
}

__global__ __launch_bounds__(kBlock) void k_make_extended_source(const double radius, const double divergence,
                                                                 const int64_t n_points, const int64_t per,
                                                                 const ArtDetectorDesc rs, const int64_t first,
                                                                 const int64_t n, const ArtBundleView out) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    art::Ray r;
    art::source_ray_extended(radius, divergence, n_points, per, rs.rot, rs.centre, first + i, r);
    store_ray(out, i, r);
    out.alive[i] = 1;
  }
}

// multi-GPU exchange (include/art_hip.h): pack statistics + a sample of the read-out; fold the gathered statistics
__global__ __launch_bounds__(kBlock) void k_exchange_pack(const double* stats, const double* X, const double* Y,
                                                          const double* opl, const uint8_t* alive, const int64_t* slots,
                                                          const int64_t k, double* send) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j < kReadoutSlots) send[j] = stats[j];
  if (j < k) {
    const int64_t s = slots[j];
    double* o = send + kReadoutSlots + 4 * j;
    o[0] = X[s]; o[1] = Y[s]; o[2] = opl[s]; o[3] = alive[s] ? 1.0 : 0.0;
  }
}

__global__ void k_exchange_fold(const double* recv, const int world, const int64_t stride, double* out) {
  const int s = threadIdx.x;
  if (s >= kReadoutSlots) return;
  const bool is_min = (s == 2 || s == 4 || s == 12), is_max = (s == 3 || s == 5 || s == 13);
  double v = recv[s];
  for (int r = 1; r < world; ++r) {       // rank order: deterministic
    const double w = recv[(int64_t)r * stride + s];
    v = is_min ? fmin(v, w) : (is_max ? fmax(v, w) : v + w);
  }
  out[s] = v;
}

// rays per launch: the hardware limit, or less when ART_MAX_RAYS_PER_LAUNCH is set (lets tests cover the chunking)
int64_t max_rays_per_launch() {
  const char* v = getenv("ART_MAX_RAYS_PER_LAUNCH");
  if (v) {
    const long long x = atoll(v);
    if (x >= 64 && x < kMaxRaysPerLaunchHw) return (int64_t)x;
  }
  return kMaxRaysPerLaunchHw;
}

// slot-offset copy of a view (launch chunking: one launch addresses at most max_rays_per_launch() slots)
ArtBundleView view_at(const ArtBundleView& v, int64_t off) {
  ArtBundleView r = v;
  if (v.alive == nullptr) return r;  // absent history view stays absent
  r.ox += off; r.oy += off; r.oz += off; r.dx += off; r.dy += off; r.dz += off;
  r.path += off; r.incidence += off; r.alive += off;
  return r;
}

bool view_ok(const ArtBundleView* v) {
  return v && v->ox && v->oy && v->oz && v->dx && v->dy && v->dz && v->path && v->incidence && v->alive;
}

int check_elem(const ArtElementDesc* e) {
  if (!e) return fail(ART_ERR_BAD_ARG, "element descriptor is NULL");
  if (e->kind < 0 || e->kind >= ART_NUM_KINDS) return fail(ART_ERR_BAD_ARG, "unknown optic kind");
  if (e->support_kind < 0 || e->support_kind > ART_SUP_RECTRECTHOLE) return fail(ART_ERR_BAD_ARG, "unknown support kind");
  if (e->n_defects < 0 || e->n_defects > ART_MAX_DEFECTS) return fail(ART_ERR_UNSUPPORTED, "too many defects on one mirror");
#ifdef ART_ZERN_LDS
  if (e->n_defects > 4) return fail(ART_ERR_UNSUPPORTED, "ART_ZERN_LDS comparison build: at most 4 Zernike tables per element");
#endif
  if (e->n_defects > 0 && e->kind == ART_MASK) return fail(ART_ERR_BAD_ARG, "a mask cannot carry defects");
  if (e->n_defects > 0 && !e->zern) return fail(ART_ERR_BAD_ARG, "n_defects > 0 but zern table is NULL");
  if (e->n_grid < 0 || e->n_grid > ART_MAX_DEFECTS) return fail(ART_ERR_UNSUPPORTED, "too many gridded defects on one mirror");
  if (e->n_grid > 0 && e->kind == ART_MASK) return fail(ART_ERR_BAD_ARG, "a mask cannot carry defects");
  if (e->n_grid > 0 && !e->grid) return fail(ART_ERR_BAD_ARG, "n_grid > 0 but grid table is NULL");
  if ((e->flags & ART_FLAG_ZERN_RECURRENCE) && e->n_defects == 0) return fail(ART_ERR_BAD_ARG, "recurrence flag without Zernike tables");
  return ART_OK;
}

template <int KIND>
void launch_element(const ArtElementDesc& e, const ArtBundleView& in, const ArtBundleView& out, int64_t n,
                    hipStream_t s) {
  ElemArg ea;
  ea.e[0] = e;
  const int xm = xcd_map();
  if (e.n_defects > 0 || e.n_grid > 0)
    hipLaunchKernelGGL((k_trace_element<KIND, true>), dim3(kDefectLoop ? grid_for(n) : grid_stream_mapped(n, xm)),
                       dim3(kBlock), 0, s, ea, in, out, n, kDefectLoop ? 0 : xm);
  else
    hipLaunchKernelGGL((k_trace_element<KIND, false>), dim3(grid_stream_mapped(n, xm)), dim3(kBlock), 0, s, ea, in, out, n,
                       xm);
}

}  // namespace

// =================================================================================================== C ABI
extern "C" {

int art_abi_version(void) { return ART_ABI_VERSION; }

const char* art_last_error(void) { return g_err; }

int art_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail_hip(e, "hipGetDeviceCount");
  }
  int good = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++good;
  }
  return good;
}

int art_trace_element(const ArtElementDesc* e, const ArtBundleView* in, const ArtBundleView* out, int64_t n,
                      void* stream) {
  int rc = check_elem(e);
  if (rc) return rc;
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  if (n == 0) return ART_OK;  // an empty bundle has no arrays to point to
  if (!view_ok(in) || !view_ok(out)) return fail(ART_ERR_BAD_ARG, "bundle view has a NULL array");
  hipStream_t s = (hipStream_t)stream;
  ArtElementDesc ec = *e;
  art::prepare_element(ec);
  const int64_t chunk = max_rays_per_launch();
  for (int64_t off = 0; off < n; off += chunk) {
    const int64_t m = (n - off < chunk) ? n - off : chunk;
    const ArtBundleView vi = view_at(*in, off), vo = view_at(*out, off);
    if (ec.flags & ART_FLAG_ZERN_RECURRENCE) {
      ElemArg ea;
      ea.e[0] = ec;
      hipLaunchKernelGGL(k_trace_element_zrec, dim3(grid_stream(m)), dim3(kBlock), 0, s, ea, vi, vo, m);
      continue;
    }
    switch (ec.kind) {
      case ART_PLANE: launch_element<ART_PLANE>(ec, vi, vo, m, s); break;
      case ART_SPHERE: launch_element<ART_SPHERE>(ec, vi, vo, m, s); break;
      case ART_PARABOLA: launch_element<ART_PARABOLA>(ec, vi, vo, m, s); break;
      case ART_TORUS: launch_element<ART_TORUS>(ec, vi, vo, m, s); break;
      case ART_ELLIPSOID: launch_element<ART_ELLIPSOID>(ec, vi, vo, m, s); break;
      case ART_CYLINDER: launch_element<ART_CYLINDER>(ec, vi, vo, m, s); break;
      default: launch_element<ART_MASK>(ec, vi, vo, m, s); break;
    }
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_trace_element launch");
  return ART_OK;
}

// launch one segment (<= 8 elements) of one or many chains
namespace {
#ifdef ART_ZERN_LDS
inline size_t zern_lds_bytes(const ChainArgs& a) {
  size_t d = 0;
  for (int k = 0; k < a.n_elems; ++k) d += (size_t)a.e[k].n_defects;
  return d * ART_ZERN_STRIDE * sizeof(double);
}
#endif
// Workgroups per launch of the separate read-out: a persistent grid of kReadoutBlocks (2048).  The scratch area has room
// for 8 x that; ART_READOUT_BLOCKS=<n> uses it (measured at 1e7 rays, tools/r02_exp21.sh: 2048 -> 157-160 us, 4096 ->
// 161, 8192 -> 161, 16384 -> 170-177: here the per-workgroup reduction outweighs what shorter-lived workgroups gain).
inline int64_t readout_block_cap(int64_t launches) {
  static const int64_t env = [] {
    const char* e = getenv("ART_READOUT_BLOCKS");
    const long x = e ? atol(e) : 0;
    return (int64_t)(x < 1 ? 0 : x);
  }();
  const int64_t room = (int64_t)8 * kReadoutBlocks / (launches < 1 ? 1 : launches);
  return (env > 0) ? (env < room ? env : room) : (kReadoutBlocks < room ? kReadoutBlocks : room);
}
// register budget of the fused kernel in waves per SIMD: 5 (91 VGPRs with the LDS store path).  A 6-wave build (80
// VGPRs) is kept as ART_CHAIN_WAVES=6 for experiments: with 8-byte stores it fitted without spills and measured 3 %
// faster on relay4 without the read-out tail, the same with it, C2 +1 %, C4 13 % slower (tools/r02_exp16.sh); with the
// LDS store path it spills 5 dwords.
inline int chain_waves() {
  const char* wv = getenv("ART_CHAIN_WAVES");
  return wv ? atoi(wv) : 5;
}
// Which body traces a chain without defects: chain_body (one ray per lane, outputs regrouped through LDS) or chain_body2
// (two rays per lane, no staging).  Measured A/B inside one process (tools/ab_kernel.py, profiles/r03_experiments.md): the
// two-ray body is 4-7 % faster on chains behind a MASK (C2, C3: a third to a half of the slots are dead -- it skips a
// dead pair with one branch and two dropped offsets, where the one-ray body still stages, synchronises and issues the
// workgroup's stores) and -3 ... +9 % elsewhere, box-dependent; so it is the default exactly where a mask is part of the
// launch -- and in a one-segment scene whose chains share their input (XCD-grouped grid, scene_wg): -8 % without a mask,
// -9 % behind one, where a lone chain (relay4) is 7 % SLOWER with it (profiles/r05_experiments.md batch s).
// ART_CHAIN_RPL=1|2 overrides (read at every call: the A/B tool alternates the variants inside one process).
// (One getenv + atoi per launch, ~0.1 us, on purpose: a cached value could not be alternated by the A/B tool.)
inline int chain_rpl(const bool has_mask) {
  const char* e = getenv("ART_CHAIN_RPL");
  const int v = e ? atoi(e) : 0;
  return (v == 1 || v == 2) ? v : (has_mask ? 2 : 1);
}
// Grid shape of a scene launch whose chains share their input: XCD-grouped by default (scene_wg).  -> 0 tile-major, 1
// chain-interleaved 2-D grid, 4 XCD-grouped.  ART_SCENE_ORDER=tile|chain|xcd forces one shape for every scene (read per
// launch: the A/B tools alternate them inside one process).
inline int scene_order(const bool shared_in) {
  const char* e = getenv("ART_SCENE_ORDER");
  if (e && e[0] == 't') return 0;
  if (e && e[0] == 'c') return 1;
  if (e && e[0] == 'x') return 4;
  return shared_in ? 4 : 0;
}
// ... and whether that shared input is loaded with the default cache policy instead of non-temporal loads.  XCD-grouped:
// always -- the C - 1 re-reads of a tile follow within microseconds in the same L2 (profiles/r05_experiments.md batch r: 10
// chains x 1e7 rays 3.167 against 3.280 ms, 11 x 1e6 0.299 against 0.314).  In the interleaved 2-D grid the re-reads come
// from the memory-side cache: worth it while the input (57 B per slot) fits its 256 MB -- 11 chains x 1e6 rays -10 %, 10
// chains x 4e6 rays (228 MB) -15 %, but 10 chains x 1e7 rays (570 MB) +6 ... +8 % (profiles/r04_experiments.md, batch 8).
// ART_SCENE_KEEP=0|1 overrides (A/B).
inline bool scene_keep(const int64_t n, const bool xcd_grouped) {
  const char* e = getenv("ART_SCENE_KEEP");
  if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
  return xcd_grouped || n * 57 <= ((int64_t)256 << 20);
}
// ART_CHAIN_DYN_LDS=<bytes>: unused dynamic LDS per workgroup of the fused kernel, i.e. FEWER resident workgroups per CU
// (20 KB static + 20480 -> 4, + 33000 -> 3).  An experiment knob: the bare access pattern gains 3-7 % of bandwidth with 2-3
// instead of 8 workgroups per CU (tools/stream_floor.hip); the kernel needs its waves to hide latency.
inline size_t chain_dyn_lds() {
  static const size_t v = [] {
    const char* e = getenv("ART_CHAIN_DYN_LDS");
    const long x = e ? atol(e) : 0;
    return (size_t)(x < 0 ? 0 : (x > 100000 ? 100000 : x));
  }();
  return v;
}
// (the fused kernel WITH defects stays at 4 waves: 112 VGPRs without spills; at 5 waves it spills 15 dwords and
// measured 0.335 instead of 0.31 ms per 1e7 rays on C5, tools/r02_exp17.sh)

}  // namespace

static int trace_chain_impl(const ArtElementDesc* elems, int32_t n_elems, const ArtBundleView* in,
                            const ArtBundleView* outs, const ArtChainReadout* ro, int64_t n, void* stream) {
  if (!elems || !outs || n_elems <= 0) return fail(ART_ERR_BAD_ARG, "empty chain");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  for (int k = 0; k < n_elems; ++k) {
    int rc = check_elem(&elems[k]);
    if (rc) return rc;
    if (elems[k].flags & ART_FLAG_ZERN_RECURRENCE)
      return fail(ART_ERR_UNSUPPORTED, "an element carries Zernike tables in the recurrence layout: trace it with art_trace_element");
  }
  hipStream_t s = (hipStream_t)stream;
  if (ro) {
    if (!art::readout_ok(*ro)) return fail(ART_ERR_BAD_ARG, "read-out: scratch/out24 missing, X/Y/opl partially NULL, or outputs / lite with sums");
    if (n > max_rays_per_launch()) return fail(ART_ERR_UNSUPPORTED, "fused read-out: more rays than one launch covers");
#ifdef ART_ZERN_LDS
    return fail(ART_ERR_UNSUPPORTED, "ART_ZERN_LDS comparison build: no fused read-out");
#endif
  }
  if (n == 0) {
    if (ro) {   // nothing to trace: the statistics are the reduction identities
      if (ro->sums) launch_sums_fold_one(ro->scratch, ro->out24, 0, s);
      else launch_fold_one(ro->scratch, ro->out24, 0, s);
      hipError_t e0 = hipGetLastError();
      if (e0 != hipSuccess) return fail_hip(e0, "art_trace_chain_readout launch");
    }
    return ART_OK;  // an empty bundle has no arrays to point to
  }
  if (!view_ok(in)) return fail(ART_ERR_BAD_ARG, "input bundle view has a NULL array");
  for (int k = 0; k < n_elems; ++k)
    if (outs[k].alive != nullptr && !view_ok(&outs[k])) return fail(ART_ERR_BAD_ARG, "history view partially NULL");
  if (!view_ok(&outs[n_elems - 1])) return fail(ART_ERR_BAD_ARG, "the last output view is mandatory");
  // a chain of ONE element is the per-element kernel's job: compiled for the optic's kind, it needs fewer registers
  // than the fused kernel's run-time dispatch (5 instead of 4 waves per SIMD with defects) -- same results
  if (n_elems == 1 && !ro) return art_trace_element(&elems[0], in, &outs[0], n, stream);
  for (int k0 = 0; k0 < n_elems; k0 += kChainMax) {
    const int m = (n_elems - k0 < kChainMax) ? n_elems - k0 : kChainMax;
    // the chunk's last bundle is the next chunk's input: it must exist
    if (!view_ok(&outs[k0 + m - 1])) return fail(ART_ERR_BAD_ARG, "chains longer than 8 need a view every 8th element");
  }
  const int waves = chain_waves();
  const int64_t chunk = max_rays_per_launch();
  for (int64_t off = 0; off < n; off += chunk) {
    const int64_t cnt = (n - off < chunk) ? n - off : chunk;
    ArtBundleView cur = view_at(*in, off);
    for (int k0 = 0; k0 < n_elems; k0 += kChainMax) {
      ChainArgs a;
      memset(&a, 0, sizeof(a));
      const int m = (n_elems - k0 < kChainMax) ? n_elems - k0 : kChainMax;
      a.n_elems = m;
      a.in = cur;
      for (int k = 0; k < m; ++k) {
        a.e[k] = elems[k0 + k];
        art::prepare_element(a.e[k]);
        a.out[k] = view_at(outs[k0 + k], off);
        if (a.e[k].n_defects > 0 || a.e[k].n_grid > 0) a.flags |= art::kFlagDefects;
      }
      const bool tail = ro && k0 + m == n_elems;       // the read-out rides on the chain's last fused launch
      if (tail) {
        a.ro = *ro;
        a.flags |= art::kFlagReadout;
      }
      size_t lds = 0;
#ifdef ART_ZERN_LDS
      lds = zern_lds_bytes(a);
      if (lds > 64 * 1024) return fail(ART_ERR_UNSUPPORTED, "ART_ZERN_LDS build: Zernike tables of one fused launch exceed 64 KiB");
#endif
      const int xm = xcd_map();
      bool has_mask = false;
      for (int k = 0; k < m; ++k) has_mask = has_mask || a.e[k].kind == ART_MASK;
      // (chains WITH defects keep the one-ray body: the two-ray one needs 133 VGPRs = 3 waves per SIMD there and measured
      // +1.6 % on C5 with the read-out, the same without; profiles/r03_experiments.md)
      const bool two = !(a.flags & art::kFlagDefects) && chain_rpl(has_mask) == 2;
      const dim3 g(grid_stream_mapped(two ? (cnt + 1) / 2 : cnt, xm)), b(kBlock);
      const int sw = (!kDefectLoop && m == 1 && special_kind(a.e[0].kind)) ? special_waves() : 0;
      if ((a.flags & art::kFlagDefects) && sw == 5)
        launch_chain1<5>(a.e[0].kind, dim3(grid_stream_mapped(cnt, xm)), s, a, cnt, xm);
      else if ((a.flags & art::kFlagDefects) && sw == 4)
        launch_chain1<4>(a.e[0].kind, dim3(grid_stream_mapped(cnt, xm)), s, a, cnt, xm);
      else if (a.flags & art::kFlagDefects)
        hipLaunchKernelGGL((k_trace_chain<true, 4>), dim3(kDefectLoop ? grid_for(cnt) : grid_stream_mapped(cnt, xm)), b, lds,
                           s, a, cnt, kDefectLoop ? 0 : xm);
      else if (two)       // 107 VGPRs: 4 waves per SIMD (3 and 5 measured the same or worse, tools/ab_kernel.py)
        hipLaunchKernelGGL((k_trace_chain2<false, 4>), g, b, chain_dyn_lds(), s, a, cnt, xm);
      else if (waves == 6)
        hipLaunchKernelGGL((k_trace_chain<false, 6>), g, b, chain_dyn_lds(), s, a, cnt, xm);
      else
        hipLaunchKernelGGL((k_trace_chain<false, 5>), g, b, chain_dyn_lds(), s, a, cnt, xm);
      if (tail && ro->sums)
        launch_sums_fold_one(ro->scratch, ro->out24, analysis_tiles(cnt), s);
      else if (tail)
        launch_fold_one(ro->scratch, ro->out24, (int64_t)g.x, s);
      cur = a.out[m - 1];
    }
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_trace_chain launch");
  return ART_OK;
}

int art_trace_chain(const ArtElementDesc* elems, int32_t n_elems, const ArtBundleView* in, const ArtBundleView* outs,
                    int64_t n, void* stream) {
  return trace_chain_impl(elems, n_elems, in, outs, nullptr, n, stream);
}

int64_t art_chain_readout_scratch_doubles(int64_t n) {
  if (n < 0) n = 0;
  if (n > kMaxRaysPerLaunchHw) n = kMaxRaysPerLaunchHw;
  // one 24-slot partial per WORKGROUP of the fused launch (its grid may be rounded up by the tile mapping) + the chunk
  // totals of the two-stage fold behind them
  return (((n + kBlock - 1) / kBlock + 1024) + kFoldChunks) * kReadoutSlots;
}

int art_trace_chain_readout(const ArtElementDesc* elems, int32_t n_elems, const ArtBundleView* in,
                            const ArtBundleView* outs, const ArtChainReadout* readout, int64_t n, void* stream) {
  if (!readout) return fail(ART_ERR_BAD_ARG, "read-out descriptor is NULL");
  return trace_chain_impl(elems, n_elems, in, outs, readout, n, stream);
}

int64_t art_scene_bytes(int32_t n_chains, int32_t n_elems) {
  if (n_chains <= 0 || n_elems <= 0) return 0;
  return art::scene_bytes(n_chains, n_elems);
}

int art_scene_pack(const ArtElementDesc* elems, int32_t n_chains, int32_t n_elems, const ArtBundleView* ins,
                   const ArtBundleView* outs, const ArtChainReadout* readouts, void* image) {
  const char* msg = "";
  const int rc = art::scene_pack(elems, n_chains, n_elems, ins, outs, readouts, image, &msg);
  if (rc < 0) return fail(rc, msg);
  return rc;
}

int art_trace_scene(const void* image_dev, const void* image_host, int64_t n, void* stream) {
  if (!image_dev || !image_host) return fail(ART_ERR_BAD_ARG, "scene image is NULL");
  // Counts and flags are read from the header art_scene_pack wrote, not taken from the caller: a read-out bit without
  // read-out descriptors in the table would make the tail dereference NULL on the device, a wrong defect bit select the
  // wrong kernel body.
  art::SceneHeader h;
  memcpy(&h, image_host, sizeof(h));
  if (h.magic != art::kSceneMagic) return fail(ART_ERR_BAD_ARG, "host image was not written by art_scene_pack");
  const int32_t n_chains = h.n_chains, n_elems = h.n_elems, flags = h.flags;
  if (n_chains <= 0 || n_chains > 65535 || n_elems <= 0 || h.n_segments != art::scene_segments(n_elems) ||
      (flags & ~(art::kFlagDefects | art::kFlagReadout | art::kFlagMask | art::kFlagSharedIn | art::kFlagSums)))
    return fail(ART_ERR_BAD_ARG, "scene header is corrupt");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
#ifdef ART_ZERN_LDS
  if (flags & 1) return fail(ART_ERR_UNSUPPORTED, "ART_ZERN_LDS build: scenes with defects go through art_trace_chain");
#endif
  if ((flags & art::kFlagReadout) && n > max_rays_per_launch())
    return fail(ART_ERR_UNSUPPORTED, "fused read-out: more rays than one launch covers");
  hipStream_t s = (hipStream_t)stream;
  const ChainArgs* tab = art::scene_table(image_dev);
  const int S = art::scene_segments(n_elems);
  if (n == 0) {
    if (flags & art::kFlagSums)
      launch_sums_fold_scene(tab + (int64_t)(S - 1) * n_chains, n_chains, 0, s);
    else if (flags & art::kFlagReadout)
      launch_fold_scene(tab + (int64_t)(S - 1) * n_chains, n_chains, 0, s);
    return ART_OK;
  }
  const int waves = chain_waves();
  // one-element scenes with defects whose chains all carry the same simple optic: the body compiled for that kind
  int kind1 = -1, special1 = 0;
  if ((flags & 1) && n_elems == 1) {
    const ChainArgs* ht = art::scene_table(image_host);
    kind1 = ht[0].e[0].kind;
    for (int c = 1; c < n_chains; ++c) kind1 = (ht[c].e[0].kind == kind1) ? kind1 : -1;
    special1 = special_kind(kind1) ? special_waves() : 0;
  }
  const int64_t chunk = max_rays_per_launch();
  for (int64_t off = 0; off < n; off += chunk) {
    const int64_t cnt = (n - off < chunk) ? n - off : chunk;
    const int xm = xcd_map();
    const bool two = !(flags & 1) && chain_rpl((flags & art::kFlagMask) != 0 ||
                                               ((flags & art::kFlagSharedIn) != 0 && n_chains > 1 && S == 1)) == 2;
    const int tiles = grid_stream_mapped(two ? (cnt + 1) / 2 : cnt, xm);
    for (int sg = 0; sg < S; ++sg) {
      const ChainArgs* seg = tab + (int64_t)sg * n_chains;
      // chains that share their input (first segment only: later segments read their own hand-over bundles) are
      // interleaved tile by tile (see k_trace_scene); ART_SCENE_ORDER=tile|chain overrides (A/B)
      // bit 0: interleaved grid; bit 1: the shared input is loaded with the default cache policy (scene_keep)
      int tr = scene_order((flags & art::kFlagSharedIn) != 0 && sg == 0 && n_chains > 1);
      const int tiles8 = (tiles + 7) / 8 * 8;       // (the padding workgroups leave at once: scene_wg)
      // (a grid dimension holds fewer than 2^32 work-items: 2^24 workgroups of 256)
      if (tr == 4 && (int64_t)tiles8 * n_chains >= ((int64_t)1 << 24)) tr = 1;
      if (tr == 1 && tiles > 65535) tr = 0;
      if (tr && scene_keep(cnt, (tr & 4) != 0)) tr |= 2;
      if (tr & 4) tr |= n_chains << 8;
      const dim3 g = (tr & 4) ? dim3((unsigned)(tiles8 * n_chains)) : ((tr & 1) ? dim3(n_chains, tiles) : dim3(tiles, n_chains)),
                 b(kBlock);
      const int xarg = (tr & 4) ? tiles : xm;      // (XCD-grouped: the true tile count travels in the mapping's parameter)
      if ((flags & 1) && special1 == 5)
        launch_scene1<5>(kind1, g, s, seg, off, cnt, xarg, tr);
      else if ((flags & 1) && special1 == 4)
        launch_scene1<4>(kind1, g, s, seg, off, cnt, xarg, tr);
      else if (flags & 1)
        hipLaunchKernelGGL((k_trace_scene<true, 4>), g, b, 0, s, seg, off, cnt, xarg, tr);
      else if (two)
        hipLaunchKernelGGL((k_trace_scene2<false, 4>), g, b, 0, s, seg, off, cnt, xarg, tr);
      else if (waves == 6)
        hipLaunchKernelGGL((k_trace_scene<false, 6>), g, b, 0, s, seg, off, cnt, xarg, tr);
      else
        hipLaunchKernelGGL((k_trace_scene<false, 5>), g, b, 0, s, seg, off, cnt, xarg, tr);
      if ((flags & art::kFlagSums) && sg == S - 1)
        launch_sums_fold_scene(seg, n_chains, analysis_tiles(cnt), s);
      else if ((flags & art::kFlagReadout) && sg == S - 1)
        launch_fold_scene(seg, n_chains, (int64_t)tiles, s);
    }
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_trace_scene launch");
  return ART_OK;
}

int art_pack_rays(const double* points, const double* vectors, const double* path0, int64_t n, const ArtBundleView* out,
                  void* stream) {
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  if (n == 0) return ART_OK;
  if (!points || !vectors || !view_ok(out)) return fail(ART_ERR_BAD_ARG, "NULL argument");
  hipLaunchKernelGGL(k_pack_rays, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, points, vectors, path0, n,
                     *out);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_pack_rays launch");
  return ART_OK;
}

int art_transform_bundle(const double M[9], const double T[3], int32_t rotate_points, const ArtBundleView* in,
                         const ArtBundleView* out, int64_t n, void* stream) {
  if (!M || !T) return fail(ART_ERR_BAD_ARG, "NULL matrix or translation");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  if (n == 0) return ART_OK;
  if (!view_ok(in) || !view_ok(out)) return fail(ART_ERR_BAD_ARG, "bundle view has a NULL array");
  ArtDetectorDesc mt;
  memset(&mt, 0, sizeof(mt));
  memcpy(mt.rot, M, 9 * sizeof(double));
  memcpy(mt.centre, T, 3 * sizeof(double));
  hipLaunchKernelGGL(k_transform, dim3(grid_stream(n)), dim3(kBlock), 0, (hipStream_t)stream, mt, (int)rotate_points, *in,
                     *out, n);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_transform_bundle launch");
  return ART_OK;
}

int art_detector(const ArtDetectorDesc* d, const ArtBundleView* b, int64_t n, double* p3x, double* p3y, double* p3z,
                 double* X, double* Y, double* opl, void* stream) {
  if (!d) return fail(ART_ERR_BAD_ARG, "detector descriptor is NULL");
  if (n == 0) return ART_OK;
  if (!view_ok(b)) return fail(ART_ERR_BAD_ARG, "bundle view has a NULL array");
  if ((p3x || p3y || p3z) && !(p3x && p3y && p3z)) return fail(ART_ERR_BAD_ARG, "p3x/p3y/p3z must be all set or all NULL");
  if ((X || Y) && !(X && Y)) return fail(ART_ERR_BAD_ARG, "X/Y must be both set or both NULL");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  if (n == 0) return ART_OK;
  hipLaunchKernelGGL(k_detector, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, *d, *b, n, p3x, p3y, p3z, X,
                     Y, opl);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_detector launch");
  return ART_OK;
}

int art_detector_readout(const ArtDetectorDesc* d, const ArtBundleView* b, const double* w, int64_t n, double cx,
                         double cy, double co, double* p3x, double* p3y, double* p3z, double* X, double* Y,
                         double* opl, double* scratch, double* out24, void* stream) {
  if (!d || !scratch || !out24) return fail(ART_ERR_BAD_ARG, "descriptor/scratch/out24 must not be NULL");
  if ((p3x || p3y || p3z) && !(p3x && p3y && p3z)) return fail(ART_ERR_BAD_ARG, "p3x/p3y/p3z must be all set or all NULL");
  if ((X || Y) && !(X && Y)) return fail(ART_ERR_BAD_ARG, "X/Y must be both set or both NULL");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    // an empty shard must not pollute a cross-rank fold: the final kernel over zero partials writes the reduction
    // identities (0 for sums, +inf / -inf for the min / max slots 2-5 and 12-13)
    hipLaunchKernelGGL(k_readout_final, dim3(kReadoutSlots), dim3(kBlock), 0, s, scratch, 0, out24);
    hipError_t e0 = hipGetLastError();
    if (e0 != hipSuccess) return fail_hip(e0, "art_detector_readout launch");
    return ART_OK;
  }
  if (!view_ok(b)) return fail(ART_ERR_BAD_ARG, "bundle view has a NULL array");
  // one launch per <= 2^28 rays (32-bit buffer offsets); every launch leaves one partial per workgroup, all of
  // them folded by the final kernel: scratch holds up to 8 launches x kReadoutBlocks workgroups x 24 doubles
  const int64_t chunk = max_rays_per_launch();
  const int64_t launches = (n + chunk - 1) / chunk;
  if (launches > 8) return fail(ART_ERR_UNSUPPORTED, "more than 2^31 rays in one read-out");
  int nb_total = 0;
  for (int64_t off = 0; off < n; off += chunk) {
    const int64_t m = (n - off < chunk) ? n - off : chunk;
    const int64_t want = ((m + 1) / 2 + kBlock - 1) / kBlock;      // two slots per thread
    const int64_t cap = readout_block_cap(launches);
    const int nb = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    const ArtBundleView v = view_at(*b, off);
    hipLaunchKernelGGL(k_detector_readout, dim3(nb), dim3(kBlock), 0, s, *d, v, w ? w + off : nullptr, m, cx, cy, co,
                       p3x ? p3x + off : nullptr, p3y ? p3y + off : nullptr, p3z ? p3z + off : nullptr,
                       X ? X + off : nullptr, Y ? Y + off : nullptr, opl ? opl + off : nullptr,
                       scratch + (int64_t)nb_total * kReadoutSlots);
    nb_total += nb;
  }
  hipLaunchKernelGGL(k_readout_final, dim3(kReadoutSlots), dim3(kBlock), 0, s, scratch, nb_total, out24);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_detector_readout launch");
  return ART_OK;
}

int art_detector_scan_moments(const ArtDetectorDesc* d, const ArtBundleView* b, const double* w, int64_t n, double co,
                              double span, double* scratch, double* out32, void* stream) {
  if (!d || !scratch || !out32) return fail(ART_ERR_BAD_ARG, "descriptor/scratch/out32 must not be NULL");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    hipError_t e0 = hipMemsetAsync(out32, 0, kScanSlots * sizeof(double), s);
    if (e0 != hipSuccess) return fail_hip(e0, "hipMemsetAsync");
    return ART_OK;
  }
  if (!view_ok(b)) return fail(ART_ERR_BAD_ARG, "bundle view has a NULL array");
  int64_t nbk = (n + kBlock - 1) / kBlock;
  const int nb = (int)(nbk > kRedBlocks ? kRedBlocks : nbk);
  if (w) hipLaunchKernelGGL(k_scan_moments_partial<true>, dim3(nb), dim3(kBlock), 0, s, *d, *b, w, n, co, span, scratch);
  else hipLaunchKernelGGL(k_scan_moments_partial<false>, dim3(nb), dim3(kBlock), 0, s, *d, *b, w, n, co, span, scratch);
  hipLaunchKernelGGL(k_scan_moments_final, dim3(kScanSlots), dim3(kBlock), 0, s, scratch, nb, out32);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_detector_scan_moments launch");
  return ART_OK;
}

int64_t art_reduce_scratch_doubles(void) { return (int64_t)8 * kReadoutBlocks * kReadoutSlots + 64; }

int art_detector_stats(const uint8_t* alive, const double* X, const double* Y, const double* opl, const double* w,
                       int64_t n, double* scratch, double* out16, void* stream) {
  if (!alive || !scratch || !out16) return fail(ART_ERR_BAD_ARG, "alive/scratch/out16 must not be NULL");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  hipStream_t s = (hipStream_t)stream;
  int64_t b = (n + kBlock - 1) / kBlock;
  const int nb = (int)(b < 1 ? 1 : (b > kRedBlocks ? kRedBlocks : b));
  if (w) hipLaunchKernelGGL(k_stats_partial<true>, dim3(nb), dim3(kBlock), 0, s, alive, X, Y, opl, w, n, scratch);
  else hipLaunchKernelGGL(k_stats_partial<false>, dim3(nb), dim3(kBlock), 0, s, alive, X, Y, opl, w, n, scratch);
  hipLaunchKernelGGL(k_stats_final, dim3(kRedSlots), dim3(kBlock), 0, s, scratch, nb, out16);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_detector_stats launch");
  return ART_OK;
}

int art_detector_moments(const uint8_t* alive, const double* X, const double* Y, const double* opl, const double* w,
                         int64_t n, double cx, double cy, double co, double* scratch, double* out8, void* stream) {
  if (!alive || !scratch || !out8) return fail(ART_ERR_BAD_ARG, "alive/scratch/out8 must not be NULL");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  hipStream_t s = (hipStream_t)stream;
  int64_t b = (n + kBlock - 1) / kBlock;
  const int nb = (int)(b < 1 ? 1 : (b > kRedBlocks ? kRedBlocks : b));
  if (w) hipLaunchKernelGGL(k_moments_partial<true>, dim3(nb), dim3(kBlock), 0, s, alive, X, Y, opl, w, n, cx, cy, co, scratch);
  else hipLaunchKernelGGL(k_moments_partial<false>, dim3(nb), dim3(kBlock), 0, s, alive, X, Y, opl, w, n, cx, cy, co, scratch);
  hipLaunchKernelGGL(k_sums_final, dim3(kSumSlots), dim3(kBlock), 0, s, scratch, nb, out8);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_detector_moments launch");
  return ART_OK;
}

int art_bundle_sums(const ArtBundleView* bv, const double* w, int64_t n, double* scratch, double* out8, void* stream) {
  if (!view_ok(bv) || !scratch || !out8) return fail(ART_ERR_BAD_ARG, "bundle/scratch/out8 must not be NULL");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  hipStream_t s = (hipStream_t)stream;
  int64_t b = (n + kBlock - 1) / kBlock;
  const int nb = (int)(b < 1 ? 1 : (b > kRedBlocks ? kRedBlocks : b));
  if (w) hipLaunchKernelGGL(k_bundle_sums_partial<true>, dim3(nb), dim3(kBlock), 0, s, *bv, w, n, scratch);
  else hipLaunchKernelGGL(k_bundle_sums_partial<false>, dim3(nb), dim3(kBlock), 0, s, *bv, w, n, scratch);
  hipLaunchKernelGGL(k_sums_final, dim3(kSumSlots), dim3(kBlock), 0, s, scratch, nb, out8);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_bundle_sums launch");
  return ART_OK;
}

static int gaussian_impl(const ArtBundleView* bv, const double axis[3], const double* axis_sums, double fraction, int64_t n,
                         double* scratch, double* w_out, void* stream);

int art_gaussian_intensity(const ArtBundleView* bv, const double axis[3], double fraction, int64_t n, double* scratch,
                           double* w_out, void* stream) {
  if (!axis) return fail(ART_ERR_BAD_ARG, "NULL argument");
  return gaussian_impl(bv, axis, nullptr, fraction, n, scratch, w_out, stream);
}

int art_gaussian_intensity_central(const ArtBundleView* bv, double fraction, int64_t n, double* scratch, double* sums8,
                                   double* w_out, void* stream) {
  if (!view_ok(bv) || !scratch || !sums8 || !w_out) return fail(ART_ERR_BAD_ARG, "NULL argument");
  if (n <= 0) return (n == 0) ? ART_OK : fail(ART_ERR_BAD_ARG, "negative ray count");
  const int rc = art_bundle_sums(bv, nullptr, n, scratch, sums8, stream);       // the central ray's sums, left on the device
  if (rc) return rc;
  const double none[3] = {0.0, 0.0, 0.0};
  return gaussian_impl(bv, none, sums8, fraction, n, scratch, w_out, stream);
}

static int gaussian_impl(const ArtBundleView* bv, const double axis[3], const double* axis_sums, double fraction, int64_t n,
                         double* scratch, double* w_out, void* stream) {
  if (!view_ok(bv) || !axis || !scratch || !w_out) return fail(ART_ERR_BAD_ARG, "NULL argument");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  if (!(fraction > 0.0 && fraction < 1.0)) return fail(ART_ERR_BAD_ARG, "fraction must be in (0, 1)");
  if (n == 0) return ART_OK;
  hipStream_t s = (hipStream_t)stream;
  int64_t b = (n + kBlock - 1) / kBlock;
  const int nb = (int)(b < 1 ? 1 : (b > kRedBlocks ? kRedBlocks : b));
  const Axis3 ax = {axis[0], axis[1], axis[2]};
  // partials in scratch[0 .. nb*8), the two maxima right behind them
  double* maxima = scratch + (int64_t)kRedBlocks * kSumSlots;
  hipLaunchKernelGGL(k_gauss_max_partial, dim3(nb), dim3(kBlock), 0, s, *bv, ax, axis_sums, n, scratch);
  hipLaunchKernelGGL(k_gauss_max_final, dim3(1), dim3(kBlock), 0, s, scratch, nb, maxima);
  hipLaunchKernelGGL(k_gauss_weights, dim3(grid_for(n)), dim3(kBlock), 0, s, *bv, ax, axis_sums, -0.5 * log(fraction), maxima, n,
                     w_out);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_gaussian_intensity launch");
  return ART_OK;
}

int art_bundle_max_angle(const ArtBundleView* bv, const double axis[3], int64_t n, double* scratch, double* out2,
                         void* stream) {
  if (!axis || !scratch || !out2) return fail(ART_ERR_BAD_ARG, "NULL argument");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    hipError_t e0 = hipMemsetAsync(out2, 0, 2 * sizeof(double), s);
    if (e0 != hipSuccess) return fail_hip(e0, "hipMemsetAsync");
    return ART_OK;
  }
  if (!view_ok(bv)) return fail(ART_ERR_BAD_ARG, "bundle view has a NULL array");
  int64_t b = (n + kBlock - 1) / kBlock;
  const int nb = (int)(b > kRedBlocks ? kRedBlocks : b);
  const Axis3 ax = {axis[0], axis[1], axis[2]};
  double* tmp = scratch + (int64_t)kRedBlocks * kSumSlots;   // 8 doubles: the folded partials
  hipLaunchKernelGGL(k_gauss_max_partial, dim3(nb), dim3(kBlock), 0, s, *bv, ax, (const double*)nullptr, n, scratch);
  hipLaunchKernelGGL(k_gauss_max_final, dim3(1), dim3(kBlock), 0, s, scratch, nb, tmp);
  hipError_t e1 = hipMemcpyAsync(out2, tmp, 2 * sizeof(double), hipMemcpyDeviceToDevice, s);
  if (e1 != hipSuccess) return fail_hip(e1, "hipMemcpyAsync");
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_bundle_max_angle launch");
  return ART_OK;
}

int64_t art_compact_scratch_ints(int64_t n) {
  const int64_t tiles = (n + kTile - 1) / kTile;
  // int32 counts [tiles] followed by int64 offsets [tiles] (8-byte aligned: round counts up to even)
  const int64_t c = (tiles + 1) & ~1ll;
  return c + 2 * tiles + 2;
}

int art_compact(const uint8_t* alive, int64_t n, int32_t* block_counts, int64_t* idx_out, int64_t* count_out,
                void* stream) {
  if (!alive || !block_counts || !idx_out || !count_out) return fail(ART_ERR_BAD_ARG, "NULL argument");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    hipError_t e0 = hipMemsetAsync(count_out, 0, sizeof(int64_t), s);
    if (e0 != hipSuccess) return fail_hip(e0, "hipMemsetAsync");
    return ART_OK;
  }
  const int64_t tiles = (n + kTile - 1) / kTile;
  const int64_t c = (tiles + 1) & ~1ll;
  int64_t* offsets = reinterpret_cast<int64_t*>(block_counts + c);
  hipLaunchKernelGGL(k_compact_count, dim3((unsigned)tiles), dim3(kBlock), 0, s, alive, n, block_counts);
  hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, s, block_counts, tiles, count_out, offsets);
  hipLaunchKernelGGL(k_compact_scatter, dim3((unsigned)tiles), dim3(kBlock), 0, s, alive, n, offsets, idx_out);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_compact launch");
  return ART_OK;
}

int64_t art_survivor_bytes(int64_t count, int32_t dense) {
  if (count < 0) count = 0;
  const int64_t b = kSurvHeader + count * (dense ? 24 : 28);
  return (b + 15) / 16 * 16;
}

int art_pack_survivors(const uint8_t* alive, int64_t n, const double* X, const double* Y, const double* opl,
                       const int64_t* number, int64_t first, int64_t step, int32_t* scratch_ints, void* send,
                       int64_t send_bytes, void* stream) {
  if (!scratch_ints || !send) return fail(ART_ERR_BAD_ARG, "NULL argument");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  if (send_bytes < art_survivor_bytes(n, 0)) return fail(ART_ERR_BAD_ARG, "send buffer smaller than art_survivor_bytes(n, 0)");
  if (((uintptr_t)send & 7u) != 0) return fail(ART_ERR_BAD_ARG, "send buffer must be 8-byte aligned");
  if (!number && (first < 0 || step < 1 || (n > 0 && first + (n - 1) * step > INT32_MAX)))
    return fail(ART_ERR_UNSUPPORTED, "ray numbers do not fit int32");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    hipError_t e0 = hipMemsetAsync(send, 0, kSurvHeader, s);
    if (e0 != hipSuccess) return fail_hip(e0, "hipMemsetAsync");
    return ART_OK;
  }
  if (!alive || !X || !Y || !opl) return fail(ART_ERR_BAD_ARG, "NULL argument");
  const int64_t tiles = (n + kTile - 1) / kTile;
  const int64_t c = (tiles + 1) & ~1ll;
  int64_t* offsets = reinterpret_cast<int64_t*>(scratch_ints + c);
  int64_t* total = offsets + tiles;    // art_compact_scratch_ints() leaves two ints behind the offsets
  hipLaunchKernelGGL(k_compact_count, dim3((unsigned)tiles), dim3(kBlock), 0, s, alive, n, scratch_ints);
  hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, s, scratch_ints, tiles, total, offsets);
  hipLaunchKernelGGL(k_survivor_header, dim3(1), dim3(64), 0, s, total, n, number ? 0 : 1, reinterpret_cast<int64_t*>(send));
  hipLaunchKernelGGL(k_survivor_scatter, dim3((unsigned)tiles), dim3(kBlock), 0, s, alive, n, offsets, X, Y, opl, number,
                     first, step, reinterpret_cast<unsigned char*>(send));
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_pack_survivors launch");
  return ART_OK;
}

int art_make_source(int32_t kind, double size, const double rot[9], const double S[3], int64_t first, int64_t n,
                    int64_t n_total, const ArtBundleView* out, void* stream) {
  return art_make_source_strided(kind, size, rot, S, first, 1, n, n_total, out, stream);
}

int art_make_source_strided(int32_t kind, double size, const double rot[9], const double S[3], int64_t first,
                            int64_t step, int64_t n, int64_t n_total, const ArtBundleView* out, void* stream) {
  if (kind != 0 && kind != 1) return fail(ART_ERR_BAD_ARG, "source kind must be 0 (point) or 1 (plane-wave disk)");
  if (!rot || !S || !view_ok(out)) return fail(ART_ERR_BAD_ARG, "NULL argument");
  if (n < 0 || first < 0 || step < 1 || n_total <= 0 || (n > 0 && first + (n - 1) * step >= n_total))
    return fail(ART_ERR_BAD_ARG, "bad index range");
  if (n == 0) return ART_OK;
  ArtDetectorDesc rs;
  memset(&rs, 0, sizeof(rs));
  memcpy(rs.rot, rot, 9 * sizeof(double));
  memcpy(rs.centre, S, 3 * sizeof(double));
  hipLaunchKernelGGL(k_make_source, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, kind, size, rs, first, step,
                     n, n_total, *out);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_make_source launch");
  return ART_OK;
}

int art_exchange_pack(const double* stats24, const double* X, const double* Y, const double* opl, const uint8_t* alive,
                      const int64_t* slots, int64_t k, double* send, void* stream) {
  if (!stats24 || !send) return fail(ART_ERR_BAD_ARG, "stats24/send must not be NULL");
  if (k < 0 || (k > 0 && (!X || !Y || !opl || !alive || !slots))) return fail(ART_ERR_BAD_ARG, "bad sample arguments");
  const int64_t work = k > kReadoutSlots ? k : kReadoutSlots;
  hipLaunchKernelGGL(k_exchange_pack, dim3((unsigned)((work + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     stats24, X, Y, opl, alive, slots, k, send);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_exchange_pack launch");
  return ART_OK;
}

int art_exchange_fold(const double* recv, int32_t world, int64_t stride_doubles, double* stats_out24, void* stream) {
  if (!recv || !stats_out24) return fail(ART_ERR_BAD_ARG, "recv/stats_out24 must not be NULL");
  if (world < 1 || stride_doubles < kReadoutSlots) return fail(ART_ERR_BAD_ARG, "bad world size or stride");
  hipLaunchKernelGGL(k_exchange_fold, dim3(1), dim3(64), 0, (hipStream_t)stream, recv, world, stride_doubles, stats_out24);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_exchange_fold launch");
  return ART_OK;
}

int art_make_extended_source(double radius, double divergence, int64_t n_points, int64_t rays_per_point,
                             const double rot[9], const double S[3], int64_t first, int64_t n,
                             const ArtBundleView* out, void* stream) {
  if (!rot || !S || !view_ok(out)) return fail(ART_ERR_BAD_ARG, "NULL argument");
  if (n_points <= 0 || rays_per_point <= 0) return fail(ART_ERR_BAD_ARG, "n_points and rays_per_point must be > 0");
  if (n < 0 || first < 0 || first + n > n_points * rays_per_point) return fail(ART_ERR_BAD_ARG, "bad index range");
  if (n == 0) return ART_OK;
  ArtDetectorDesc rs;
  memset(&rs, 0, sizeof(rs));
  memcpy(rs.rot, rot, 9 * sizeof(double));
  memcpy(rs.centre, S, 3 * sizeof(double));
  hipLaunchKernelGGL(k_make_extended_source, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, radius,
                     divergence, n_points, rays_per_point, rs, first, n, *out);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_make_extended_source launch");
  return ART_OK;
}

int art_trace_guides(const ArtElementDesc* elems, int32_t count, double* rays, uint8_t* alive, void* stream) {
  if (!elems || !rays || !alive) return fail(ART_ERR_BAD_ARG, "NULL argument");
  if (count < 0 || count > ART_GUIDES_MAX) return fail(ART_ERR_BAD_ARG, "art_trace_guides: count must be in 0..8");
  if (count == 0) return ART_OK;
  GuideArgs ga;
  memset(&ga, 0, sizeof(ga));
  bool defects = false;
  for (int k = 0; k < count; ++k) {
    int rc = check_elem(&elems[k]);
    if (rc) return rc;
    if (elems[k].flags & ART_FLAG_ZERN_RECURRENCE)
      return fail(ART_ERR_UNSUPPORTED, "an element carries Zernike tables in the recurrence layout: trace it with art_trace_element");
    ga.e[k] = elems[k];
    art::prepare_element(ga.e[k]);
    defects = defects || ga.e[k].n_defects > 0 || ga.e[k].n_grid > 0;
  }
  hipStream_t s = (hipStream_t)stream;
  if (defects) hipLaunchKernelGGL(k_trace_guides<true>, dim3(1), dim3(64), 0, s, ga, rays, alive, (int)count);
  else hipLaunchKernelGGL(k_trace_guides<false>, dim3(1), dim3(64), 0, s, ga, rays, alive, (int)count);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_trace_guides launch");
  return ART_OK;
}

// scratch layout of art_analyse_bundles: [job][kSumRows rows x ntiles + chunk totals] | [job][kAnaSumsPad] | [job][P][kAnaMom] | [job][kAnaPlace]
static int64_t analysis_rows_stride(int64_t n) { return (int64_t)kSumRows * (analysis_tiles(n) + kFoldChunks); }
int64_t art_analysis_scratch_doubles(int32_t n_jobs, int64_t n) {
  if (n_jobs < 1) n_jobs = 1;
  if (n < 0) n = 0;
  return (int64_t)n_jobs * (analysis_rows_stride(n) + kAnaSumsPad + (int64_t)kAnaBlocks * kAnaMom + kAnaPlace);
}

int art_analyse_bundles(const ArtAnalysisJob* jobs_dev, const ArtAnalysisJob* jobs_host, int32_t n_jobs, int64_t n,
                        double* scratch, double* out, void* stream) {
  if (!jobs_dev || !jobs_host || !scratch || !out) return fail(ART_ERR_BAD_ARG, "NULL argument");
  if (n_jobs <= 0 || n_jobs > 65535) return fail(ART_ERR_BAD_ARG, "n_jobs must be in 1..65535");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  if (n > kMaxRaysPerLaunchHw) return fail(ART_ERR_UNSUPPORTED, "more than 2^28 rays per bundle in one analysis");
  bool all_given = true;
  for (int j = 0; j < n_jobs; ++j) {
    const ArtAnalysisJob& jb = jobs_host[j];
    if (jb.mode < ART_JOB_AUTOPLACE || jb.mode > ART_JOB_SUMS) return fail(ART_ERR_BAD_ARG, "unknown job mode");
    if (n > 0 && !view_ok(&jb.b)) return fail(ART_ERR_BAD_ARG, "a job's bundle view has a NULL array");
    if (jb.mode == ART_JOB_MANUAL) {
      // the manual detector's normal is used bit for bit (rotation_to_ez and the Kahan angle assume a unit vector)
      const double nn = jb.normal[0] * jb.normal[0] + jb.normal[1] * jb.normal[1] + jb.normal[2] * jb.normal[2];
      if (!(fabs(nn - 1.0) <= 1e-12)) return fail(ART_ERR_BAD_ARG, "a manual job's detector normal is not a unit vector");
    }
    all_given = all_given && jb.sums != nullptr;
  }
  hipStream_t s = (hipStream_t)stream;
  const int P = analysis_blocks(n);
  const int64_t ntiles = analysis_tiles(n), rstride = analysis_rows_stride(n);
  double* rows = scratch;
  double* sums = rows + (int64_t)n_jobs * rstride;
  double* mom = sums + (int64_t)n_jobs * kAnaSumsPad;
  double* place = mom + (int64_t)n_jobs * P * kAnaMom;
  const int direct = ntiles <= kFoldDirect;
  if (!all_given && n > 0) {
    for (int j0 = 0; j0 < n_jobs;) {       // one launch per run of consecutive jobs without sums of their own
      if (jobs_host[j0].sums != nullptr) { ++j0; continue; }
      int j1 = j0;
      while (j1 < n_jobs && jobs_host[j1].sums == nullptr) ++j1;
      hipLaunchKernelGGL(k_analysis_sums, dim3((unsigned)ntiles, j1 - j0), dim3(kBlock), 0, s, jobs_dev, j0, n, rows, rstride);
      if (!direct)
        hipLaunchKernelGGL(k_analysis_sums_fold1, dim3(kSumRows, kFoldChunks, j1 - j0), dim3(kFoldBlock), 0, s, jobs_dev, j0, rows,
                           rstride, ntiles);
      j0 = j1;
    }
  }
  // (n == 0: no tiles were written; a fold over zero tiles leaves the sums' identities -- all zero)
  hipLaunchKernelGGL(k_analysis_sums_fold2, dim3(kSumRows, 1, n_jobs), dim3(direct ? fold_threads(n > 0 ? ntiles : 0) : kBlock), 0, s,
                     jobs_dev, rows, rstride, n > 0 ? ntiles : 0, direct, sums);
  hipLaunchKernelGGL(k_analysis_place, dim3((n_jobs + 63) / 64), dim3(64), 0, s, jobs_dev, (int)n_jobs, sums, place, out);
  {
    // (a grid dimension holds fewer than 2^32 work-items = 2^24 workgroups: more than 16 383 jobs go in several launches)
    const int P8 = (P + 7) / 8 * 8, jmax = ((1 << 24) - 1) / P8, jm = analysis_job_major();
    for (int j0 = 0; j0 < n_jobs; j0 += jmax) {
      const int J = (n_jobs - j0 < jmax) ? n_jobs - j0 : jmax;
      hipLaunchKernelGGL(k_analysis_moments, dim3((unsigned)P8 * (unsigned)J), dim3(kBlock), 0, s, jobs_dev, n, place, out, mom, P,
                         j0, J, jm);
    }
  }
  hipLaunchKernelGGL(k_analysis_fold, dim3(kAnaMom - 1, n_jobs), dim3(kBlock), 0, s, jobs_dev, P, mom, out);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_analyse_bundles launch");
  return ART_OK;
}

int art_survivor_finish(const double* stats24, int64_t n, void* send, double* xhdr, void* stream) {
  if (!stats24 || !send) return fail(ART_ERR_BAD_ARG, "NULL argument");
  if (n < 0) return fail(ART_ERR_BAD_ARG, "negative ray count");
  if (((uintptr_t)send & 15u) != 0) return fail(ART_ERR_BAD_ARG, "send buffer must be 16-byte aligned");
  hipLaunchKernelGGL(k_survivor_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, stats24, n, reinterpret_cast<int64_t*>(send), xhdr);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_survivor_finish launch");
  return ART_OK;
}

int art_survivor_xheader(const void* send, const double* stats24, double* xhdr, void* stream) {
  if (!send || !xhdr) return fail(ART_ERR_BAD_ARG, "NULL argument");
  if (((uintptr_t)send & 7u) != 0) return fail(ART_ERR_BAD_ARG, "send buffer must be 8-byte aligned");
  hipLaunchKernelGGL(k_survivor_xheader, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<const int64_t*>(send), stats24, xhdr);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail_hip(err, "art_survivor_xheader launch");
  return ART_OK;
}

}  // extern "C"
