// art_twin.cpp -- CPU twin of the gfx950 kernels.  TEST INFRASTRUCTURE ONLY (lives under oracle/).
//
// Compiles the SAME per-ray functions as the HIP kernels (attosecondraytracing_amd/csrc/art_device.h,
// -DART_HOST_TWIN) with g++ and runs them in plain host loops over HOST arrays, so that the kernel math
// (closed-form quadrics, convex-Newton torus solver, Zernike recurrences, 3x3 frame maps) can be checked
// against the oracle and the golden vectors in a container without a GPU.  The product never loads it.
#define ART_HOST_TWIN 1
#include <vector>
#include "../../attosecondraytracing_amd/csrc/art_device.h"
#include "../../attosecondraytracing_amd/csrc/art_scene.h"

#include <string.h>

namespace {
inline void load_ray(const ArtBundleView& v, int64_t i, art::Ray& r) {
  r.ox = v.ox[i]; r.oy = v.oy[i]; r.oz = v.oz[i];
  r.dx = v.dx[i]; r.dy = v.dy[i]; r.dz = v.dz[i];
  r.path = v.path[i];
}
inline void store_ray(const ArtBundleView& v, int64_t i, const art::Ray& r) {
  v.ox[i] = r.ox; v.oy[i] = r.oy; v.oz[i] = r.oz;
  v.dx[i] = r.dx; v.dy[i] = r.dy; v.dz[i] = r.dz;
  v.path[i] = r.path;
  v.incidence[i] = r.inc;
}
}  // namespace

extern "C" {

int art_cpu_trace_element(const ArtElementDesc* e_in, const ArtBundleView* in, const ArtBundleView* out, int64_t n) {
  ArtElementDesc ec = *e_in;
  art::prepare_element(ec);
  const ArtElementDesc* e = &ec;
  for (int64_t i = 0; i < n; ++i) {
    bool ok = in->alive[i] != 0;
    art::Ray r;
    if (ok) {
      load_ray(*in, i, r);
      ok = art::trace_ray_dyn<true>(*e, e->zern, r);
    }
    if (ok) store_ray(*out, i, r);
    out->alive[i] = ok ? 1 : 0;
  }
  return 0;
}

int art_cpu_trace_chain(const ArtElementDesc* elems_in, int32_t n_elems, const ArtBundleView* in,
                        const ArtBundleView* outs, int64_t n) {
  std::vector<ArtElementDesc> elems(elems_in, elems_in + n_elems);
  for (int k = 0; k < n_elems; ++k) art::prepare_element(elems[k]);
  // rays are independent: all host cores (used by bench.py's all-cores CPU figure; the tests do not care)
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    bool ok = in->alive[i] != 0;
    art::Ray r;
    if (ok) load_ray(*in, i, r);
    for (int k = 0; k < n_elems; ++k) {
      if (ok) ok = art::trace_ray_dyn<true>(elems[k], elems[k].zern, r);
      if (outs[k].alive != nullptr) {
        if (ok) store_ray(outs[k], i, r);
        outs[k].alive[i] = ok ? 1 : 0;
      }
    }
  }
  return 0;
}

// scene table (csrc/art_scene.h): the same host-side packer as the HIP library, then one host loop per chain segment
int64_t art_cpu_scene_bytes(int32_t n_chains, int32_t n_elems) {
  if (n_chains <= 0 || n_elems <= 0) return 0;
  return art::scene_bytes(n_chains, n_elems);
}

int art_cpu_scene_pack(const ArtElementDesc* elems, int32_t n_chains, int32_t n_elems, const ArtBundleView* ins,
                       const ArtBundleView* outs, const ArtChainReadout* ros, void* image) {
  const char* msg = "";
  return art::scene_pack(elems, n_chains, n_elems, ins, outs, ros, image, &msg);
}

// fused read-out of a chain's last bundle: per-ray outputs here, the statistics are left to the caller (the test
// backend reduces them with NumPy, as it does for the separate read-out)
static void readout_tail(const ArtChainReadout& ro, const ArtBundleView& last, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    if (last.alive[i] == 0) continue;
    art::Ray r;
    load_ray(last, i, r);
    double Ix, Iy, Iz, x, y, o;
    art::detector_ray(ro.det, r, Ix, Iy, Iz, x, y, o);
    if (ro.X) { ro.X[i] = x; ro.Y[i] = y; ro.opl[i] = o; }
  }
}

int art_cpu_trace_scene(const void* image, const void* image_host, int64_t n) {
  art::SceneHeader h;
  memcpy(&h, image_host, sizeof(h));
  if (h.magic != art::kSceneMagic) return ART_ERR_BAD_ARG;
  const int n_chains = h.n_chains, n_elems = h.n_elems;
  const art::ChainArgs* tab = art::scene_table(image);
  const int S = art::scene_segments(n_elems);
  for (int sg = 0; sg < S; ++sg) {
    for (int c = 0; c < n_chains; ++c) {
      const art::ChainArgs& a = tab[(int64_t)sg * n_chains + c];
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) {
        bool ok = a.in.alive[i] != 0;
        art::Ray r;
        if (ok) load_ray(a.in, i, r);
        for (int k = 0; k < a.n_elems; ++k) {
          if (ok) ok = art::trace_ray_dyn<true>(a.e[k], a.e[k].zern, r);
          if (a.out[k].alive != nullptr) {
            if (ok) store_ray(a.out[k], i, r);
            a.out[k].alive[i] = ok ? 1 : 0;
          }
        }
      }
      if ((a.flags & art::kFlagReadout) && !a.ro.sums) readout_tail(a.ro, a.out[a.n_elems - 1], n);
    }
  }
  return 0;
}

int art_cpu_chain_readout_tail(const ArtChainReadout* ro, const ArtBundleView* last, int64_t n) {
  readout_tail(*ro, *last, n);
  return 0;
}

int art_cpu_detector(const ArtDetectorDesc* d, const ArtBundleView* b, int64_t n, double* p3x, double* p3y,
                     double* p3z, double* X, double* Y, double* opl) {
  for (int64_t i = 0; i < n; ++i) {
    if (b->alive[i] == 0) continue;
    art::Ray r;
    load_ray(*b, i, r);
    double Ix, Iy, Iz, x, y, o;
    art::detector_ray(*d, r, Ix, Iy, Iz, x, y, o);
    if (p3x) { p3x[i] = Ix; p3y[i] = Iy; p3z[i] = Iz; }
    if (X) { X[i] = x; Y[i] = y; }
    if (opl) opl[i] = o;
  }
  return 0;
}

int art_cpu_detector_scan(const ArtDetectorDesc* d, const ArtBundleView* b, int64_t n, double span, double* X,
                          double* Y, double* O, double* sx, double* sy, double* so, double* crosses) {
  for (int64_t i = 0; i < n; ++i) {
    if (b->alive[i] == 0) continue;
    art::Ray r;
    load_ray(*b, i, r);
    bool c;
    art::detector_ray_scan(*d, r, span, X[i], Y[i], O[i], sx[i], sy[i], so[i], c);
    crosses[i] = c ? 1.0 : 0.0;
  }
  return 0;
}

int art_cpu_transform_bundle(const double* M, const double* T, int32_t rotate_points, const ArtBundleView* in,
                             const ArtBundleView* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    art::Ray r;
    load_ray(*in, i, r);
    r.inc = in->incidence[i];
    double px = r.ox, py = r.oy, pz = r.oz, vx, vy, vz;
    if (rotate_points) art::mat3_apply(M, r.ox, r.oy, r.oz, px, py, pz);
    art::mat3_apply(M, r.dx, r.dy, r.dz, vx, vy, vz);
    const double inv = 1.0 / sqrt(art::dot3(vx, vy, vz, vx, vy, vz));
    r.ox = px + T[0]; r.oy = py + T[1]; r.oz = pz + T[2];
    r.dx = vx * inv; r.dy = vy * inv; r.dz = vz * inv;
    store_ray(*out, i, r);
    out->alive[i] = in->alive[i];
  }
  return 0;
}

int art_cpu_pack_rays(const double* points, const double* vectors, const double* path0, int64_t n,
                      const ArtBundleView* out) {
  for (int64_t i = 0; i < n; ++i) {
    art::Ray r;
    r.ox = points[3 * i]; r.oy = points[3 * i + 1]; r.oz = points[3 * i + 2];
    const double vx = vectors[3 * i], vy = vectors[3 * i + 1], vz = vectors[3 * i + 2];
    const double inv = 1.0 / sqrt(art::dot3(vx, vy, vz, vx, vy, vz));
    r.dx = vx * inv; r.dy = vy * inv; r.dz = vz * inv;
    r.path = path0 ? path0[i] : 0.0;
    r.inc = NAN;
    store_ray(*out, i, r);
    out->alive[i] = 1;
  }
  return 0;
}

int art_cpu_make_source(int32_t kind, double size, const double* rot, const double* S, int64_t first, int64_t step,
                        int64_t n, int64_t n_total, const ArtBundleView* out) {
  for (int64_t i = 0; i < n; ++i) {
    art::Ray r;
    art::source_ray(kind, size, rot, S, first + i * step, n_total, r);
    store_ray(*out, i, r);
    out->alive[i] = 1;
  }
  return 0;
}

int art_cpu_make_extended_source(double radius, double divergence, int64_t n_points, int64_t per, const double* rot,
                                 const double* S, int64_t first, int64_t n, const ArtBundleView* out) {
  for (int64_t i = 0; i < n; ++i) {
    art::Ray r;
    art::source_ray_extended(radius, divergence, n_points, per, rot, S, first + i, r);
    store_ray(*out, i, r);
    out->alive[i] = 1;
  }
  return 0;
}

// guide rays of the placement (art_trace_guides): ray j through element j, in place
int art_cpu_trace_guides(const ArtElementDesc* elems, int32_t count, double* rays, uint8_t* alive) {
  for (int j = 0; j < count; ++j) {
    if (alive[j] == 0) continue;
    ArtElementDesc e = elems[j];
    art::prepare_element(e);
    double* q = rays + 8 * j;
    art::Ray r = {q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]};
    const bool ok = art::trace_ray_dyn<true>(e, e.zern, r);
    if (ok) { q[0] = r.ox; q[1] = r.oy; q[2] = r.oz; q[3] = r.dx; q[4] = r.dy; q[5] = r.dz; q[6] = r.path; q[7] = r.inc; }
    alive[j] = ok ? 1 : 0;
  }
  return 0;
}

// detector placement of art_analyse_bundles (art_device.h analysis_place): out = centre(3), normal(3), rot(9), refpoint(3),
// axis(3), co
int art_cpu_analysis_place(const double* sums9, int32_t mode, double distance, const double* centre, const double* normal,
                           const double* refpoint, double* out22) {
  ArtDetectorDesc d;
  double ref[3], axis[3], co;
  art::analysis_place(sums9, mode, distance, centre, normal, refpoint, d, ref, axis, co);
  for (int k = 0; k < 3; ++k) { out22[k] = d.centre[k]; out22[3 + k] = d.normal[k]; out22[15 + k] = ref[k]; out22[18 + k] = axis[k]; }
  for (int k = 0; k < 9; ++k) out22[6 + k] = d.rot[k];
  out22[21] = co;
  return 0;
}

int art_cpu_detector_scan_kink(const ArtDetectorDesc* d, const ArtBundleView* b, int64_t n, double* X, double* Y, double* O,
                               double* sx, double* sy, double* so, double* skink) {
  for (int64_t i = 0; i < n; ++i) {
    if (b->alive[i] == 0) continue;
    art::Ray r;
    load_ray(*b, i, r);
    double un;
    const double inv_nn = 1.0 / art::dot3(d->normal[0], d->normal[1], d->normal[2], d->normal[0], d->normal[1], d->normal[2]);
    art::detector_ray_scan_kink(*d, r, inv_nn, X[i], Y[i], O[i], sx[i], sy[i], so[i], skink[i], un);
  }
  return 0;
}

// test hooks for the scalar helpers of art_device.h (tests/test_device_math.py): n values each
void art_cpu_kahan_angle_unit(const double* u, const double* v, int64_t n, double* out) {
  for (int64_t i = 0; i < n; ++i)
    out[i] = art::kahan_angle_unit(u[3 * i], u[3 * i + 1], u[3 * i + 2], v[3 * i], v[3 * i + 1], v[3 * i + 2],
                                   art::dot3(u[3 * i], u[3 * i + 1], u[3 * i + 2], v[3 * i], v[3 * i + 1], v[3 * i + 2]));
}
void art_cpu_atan01(const double* q, int64_t n, double* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = art::atan01(q[i]);
}
// prepare_element on a copy: the derived slots (art_device.h) for inspection
void art_cpu_prepare_element(const ArtElementDesc* in, ArtElementDesc* out) {
  *out = *in;
  art::prepare_element(*out);
}

}  // extern "C"
