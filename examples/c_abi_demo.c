/* A plain-C consumer of libart_hip.so: no Python, no PyTorch -- device memory from the HIP runtime, everything else
 * through include/art_hip.h.  A point source (1e6 rays, 20 mrad half-angle) at the origin looks along +x at a plane
 * mirror 500 mm away under 45 degrees; the reflected bundle (travelling along +y) is read out on a detector 300 mm
 * behind the mirror.  Known answers: every ray survives, and since a plane mirror only folds the beam, every
 * optical path to the detector plane equals the straight distance from the mirror image of the source.  Then the two
 * entry points of ABI v10: the bundle analysed with a detector placed by the library (art_analyse_bundles) and one
 * alignment ray through the mirror (art_trace_guides).
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_demo.c \
 *       -L/opt/rocm/lib -lamdhip64 attosecondraytracing_amd/libart_hip.so -lm -o build/c_abi_demo
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "art_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_ART(x) do { int r_ = (x); if (r_ != ART_OK) { fprintf(stderr, "%s: %s\n", #x, art_last_error()); return 3; } } while (0)

/* one bundle = one device block of 8 rows pitched to 512 bytes + the alive bytes */
static int alloc_bundle(int64_t n, ArtBundleView* v) {
  const int64_t pitch = ((n + 63) / 64) * 64;
  double* block;
  if (hipMalloc((void**)&block, (size_t)pitch * 8 * sizeof(double)) != hipSuccess) return 1;
  if (hipMalloc((void**)&v->alive, (size_t)n) != hipSuccess) return 1;
  v->ox = block; v->oy = block + pitch; v->oz = block + 2 * pitch;
  v->dx = block + 3 * pitch; v->dy = block + 4 * pitch; v->dz = block + 5 * pitch;
  v->path = block + 6 * pitch; v->incidence = block + 7 * pitch;
  return 0;
}

int main(void) {
  if (art_abi_version() != ART_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
  if (art_device_count() < 1) { fprintf(stderr, "no gfx950 device: %s\n", art_last_error()); return 1; }
  const int64_t n = 1000000;
  ArtBundleView src, out;
  if (alloc_bundle(n, &src) || alloc_bundle(n, &out)) { fprintf(stderr, "hipMalloc failed\n"); return 2; }

  /* source: cone about +x.  rot maps ez onto ex (a rotation about ey): rows (0,0,1), (0,1,0), (-1,0,0) */
  const double rot[9] = {0, 0, 1, 0, 1, 0, -1, 0, 0}, S[3] = {0, 0, 0};
  CHECK_ART(art_make_source(0, 0.02, rot, S, 0, n, n, &src, NULL));

  /* plane mirror at (500,0,0), normal (-1,1,0)/sqrt2, major axis (1,1,0)/sqrt2, round aperture of radius 100 mm.
   * fwd = rows (major, normal x major... ) -- the optic frame has ez = normal, ex = major axis */
  ArtElementDesc e;
  memset(&e, 0, sizeof(e));
  e.kind = ART_PLANE;
  e.support_kind = ART_SUP_ROUND;
  e.sp[0] = 100.0;
  const double s = sqrt(0.5);
  const double ex[3] = {s, s, 0}, ez[3] = {-s, s, 0};
  const double ey[3] = {ez[1] * ex[2] - ez[2] * ex[1], ez[2] * ex[0] - ez[0] * ex[2], ez[0] * ex[1] - ez[1] * ex[0]};
  for (int k = 0; k < 3; ++k) {
    e.fwd[k] = ex[k]; e.fwd[3 + k] = ey[k]; e.fwd[6 + k] = ez[k];     /* lab -> optic: rows = optic axes */
    e.bwd[3 * k] = ex[k]; e.bwd[3 * k + 1] = ey[k]; e.bwd[3 * k + 2] = ez[k];
  }
  e.pos[0] = 500.0;
  CHECK_ART(art_trace_element(&e, &src, &out, n, NULL));

  /* detector: 300 mm behind the mirror along +y, facing the beam */
  ArtDetectorDesc d;
  memset(&d, 0, sizeof(d));
  d.centre[0] = 500.0; d.centre[1] = 300.0;
  d.normal[1] = -1.0;
  const double drot[9] = {1, 0, 0, 0, 0, 1, 0, -1, 0};                 /* any rotation taking the normal to ez */
  memcpy(d.rot, drot, sizeof(drot));
  double *X, *Y, *opl, *scratch, *stats_dev;
  CHECK_HIP(hipMalloc((void**)&X, n * sizeof(double)));
  CHECK_HIP(hipMalloc((void**)&Y, n * sizeof(double)));
  CHECK_HIP(hipMalloc((void**)&opl, n * sizeof(double)));
  CHECK_HIP(hipMalloc((void**)&scratch, (size_t)art_reduce_scratch_doubles() * sizeof(double)));
  CHECK_HIP(hipMalloc((void**)&stats_dev, 24 * sizeof(double)));
  CHECK_ART(art_detector_readout(&d, &out, NULL, n, 0.0, 0.0, 800.0, NULL, NULL, NULL, X, Y, opl, scratch, stats_dev, NULL));
  double st[24];
  CHECK_HIP(hipMemcpy(st, stats_dev, sizeof(st), hipMemcpyDeviceToHost));

  /* check a few rays on the host: |image - hit on detector| must equal the optical path */
  enum { K = 5 };
  const int64_t pick[K] = {0, 1, 1234, 500000, n - 1};
  double worst = 0.0;
  for (int j = 0; j < K; ++j) {
    double x, y, o;
    CHECK_HIP(hipMemcpy(&x, X + pick[j], 8, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&y, Y + pick[j], 8, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&o, opl + pick[j], 8, hipMemcpyDeviceToHost));
    /* detector coordinates: X = lab x - 500, Y = lab z (drot); hit = (500 + x, 300, y); image of the source = (500, -500, 0) */
    const double hx = 500.0 + x - 500.0, hy = 300.0 + 500.0, hz = y;
    const double straight = sqrt(hx * hx + hy * hy + hz * hz);
    if (fabs(straight - o) > worst) worst = fabs(straight - o);
  }
  printf("rays %lld  alive %.0f  mean path %.9f mm  max |path - distance from image| %.3e mm\n", (long long)n, st[0],
         st[1] / st[0], worst);
  if (st[0] != (double)n || worst > 1e-9) { fprintf(stderr, "C_ABI_DEMO_FAILED\n"); return 4; }

  /* ABI v10, part 1 -- art_analyse_bundles: the same bundle analysed WITHOUT a given detector.  The library places one on
   * the mean ray, 300 mm from the mean hit point (Detector.autoplace), and returns the read-out moments on it.  For this
   * symmetric cone the mean direction behind the mirror is +y exactly, the mean hit point lies in the mirror plane
   * (y = x - 500; a little beyond its centre: the footprint of a cone on a tilted plane is lopsided, by 500 <a^2> = 0.05
   * mm), so the detector comes out parallel to the one above, shifted by that much, and the mean path longer by the shift. */
  ArtAnalysisJob job;
  memset(&job, 0, sizeof(job));
  job.b = out;
  job.mode = ART_JOB_AUTOPLACE;
  job.distance = 300.0;
  ArtAnalysisJob* job_dev;
  double *ana_scratch, *ana_out;
  CHECK_HIP(hipMalloc((void**)&job_dev, sizeof(job)));
  CHECK_HIP(hipMemcpy(job_dev, &job, sizeof(job), hipMemcpyHostToDevice));
  CHECK_HIP(hipMalloc((void**)&ana_scratch, (size_t)art_analysis_scratch_doubles(1, n) * sizeof(double)));
  CHECK_HIP(hipMalloc((void**)&ana_out, ART_ANALYSIS_DOUBLES * sizeof(double)));
  CHECK_ART(art_analyse_bundles(job_dev, &job, 1, n, ana_scratch, ana_out, NULL));
  double row[ART_ANALYSIS_DOUBLES];
  CHECK_HIP(hipMemcpy(row, ana_out, sizeof(row), hipMemcpyDeviceToHost));
  const double mean_opl = row[19] + row[20 + 11] / row[20];            /* co + sum(opl - co) / count */
  printf("analysis: count %.0f  detector centre (%.6f, %.6f, %.6f) normal (%.6f, %.6f, %.6f)  mean path %.9f mm  "
         "largest angle to the mean ray %.6f rad\n", row[0], row[10], row[11], row[12], row[13], row[14], row[15], mean_opl, row[55]);
  const double shift = row[11] - 300.0;
  /* (a Vogel spiral of 1e6 rays is symmetric to ~1e-8 only: the tolerances below are those of the discretisation) */
  if (row[0] != (double)n || shift < 0.0 || shift > 0.2 || fabs((row[10] - 500.0) - shift) > 1e-4 || fabs(row[12]) > 1e-4 ||
      fabs(row[13]) > 1e-6 || fabs(row[14] + 1.0) > 1e-9 || fabs(mean_opl - st[1] / st[0] - shift) > 1e-4 ||
      fabs(row[55] - 0.02) > 1e-5) {
    fprintf(stderr, "C_ABI_DEMO_FAILED (analysis)\n");
    return 5;
  }

  /* ABI v10, part 2 -- art_trace_guides: one alignment ray along +x through the same mirror: it must leave along +y
   * from (500, 0, 0) with a path of 500 mm and an incidence angle of 45 degrees. */
  const double guide[8] = {0, 0, 0, 1, 0, 0, 0, 0};
  double* guide_dev;
  uint8_t* guide_alive;
  const uint8_t one = 1;
  CHECK_HIP(hipMalloc((void**)&guide_dev, sizeof(guide)));
  CHECK_HIP(hipMalloc((void**)&guide_alive, 1));
  CHECK_HIP(hipMemcpy(guide_dev, guide, sizeof(guide), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(guide_alive, &one, 1, hipMemcpyHostToDevice));
  CHECK_ART(art_trace_guides(&e, 1, guide_dev, guide_alive, NULL));
  double g[8];
  uint8_t ga;
  CHECK_HIP(hipMemcpy(g, guide_dev, sizeof(g), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(&ga, guide_alive, 1, hipMemcpyDeviceToHost));
  printf("guide ray: alive %d  point (%.9f, %.9f, %.9f)  direction (%.9f, %.9f, %.9f)  path %.9f  incidence %.9f rad\n", ga, g[0],
         g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
  if (ga != 1 || fabs(g[0] - 500.0) > 1e-9 || fabs(g[1]) > 1e-9 || fabs(g[3]) > 1e-12 || fabs(g[4] - 1.0) > 1e-12 ||
      fabs(g[6] - 500.0) > 1e-9 || fabs(g[7] - 0.78539816339744831) > 1e-12) {
    fprintf(stderr, "C_ABI_DEMO_FAILED (guide ray)\n");
    return 6;
  }
  printf("C_ABI_DEMO_OK\n");
  return 0;
}
