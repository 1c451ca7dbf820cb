/*
 * art_hip.h -- C ABI of libart_hip.so: MI355X (gfx950) ray-bundle propagation for ART.
 *
 * The reference (mightymightys/AttosecondRaytracing, pure Python) has no FFI layer; its boundary for
 * this path is the Python call
 *     ModuleProcessing.RayTracingCalculation(source_rays, optical_elements, IgnoreDefects)
 *         ART/ModuleProcessing.py:250-313, sole call site ART/ModuleOpticalChain.py:194
 * plus the Detector read-out methods ART/ModuleDetector.py:191-279.  The entry points below are what a
 * ctypes binding on the reference side would call instead of those Python loops (INTEGRATION.md shows
 * the stub).  Conventions:
 *   - plain C: pointers, sizes, POD structs; no C++/torch types cross the boundary;
 *   - every ray array is a DEVICE pointer owned by the caller (fp64 SoA, one entry per source ray, fixed
 *     length n for the whole chain; rays that missed an element keep their slot with alive = 0);
 *   - descriptors (ArtElementDesc, ArtDetectorDesc) are HOST structs, copied into kernel arguments;
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = default);
 *   - return value 0 = ART_OK, negative = error; art_last_error() gives a thread-local message;
 *   - no mutable global state, re-entrant, current HIP device is used.
 */
#ifndef ART_HIP_H
#define ART_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ART_ABI_VERSION 11

/* error codes */
#define ART_OK 0
#define ART_ERR_BAD_ARG (-1)      /* null pointer, negative size, unknown kind                      */
#define ART_ERR_UNSUPPORTED (-2)  /* e.g. Zernike order above ART_ZERN_MAX_ORDER                    */
#define ART_ERR_HIP (-3)          /* a HIP runtime call failed (message holds hipGetErrorString)     */
#define ART_ERR_NO_DEVICE (-4)    /* no gfx950 device visible                                        */

/* optic kinds: ART/ModuleMirror.py classes + ART/ModuleMask.py                                       */
enum ArtOpticKind {
  ART_PLANE = 0,      /* MirrorPlane        ModuleMirror.py:42-113   */
  ART_SPHERE = 1,     /* MirrorSpherical    ModuleMirror.py:117-208  mp[0]=|R|                        */
  ART_PARABOLA = 2,   /* MirrorParabolic    ModuleMirror.py:212-387  mp[0]=p (semi latus rectum)      */
  ART_TORUS = 3,      /* MirrorToroidal     ModuleMirror.py:391-527  mp[0]=R major, mp[1]=r minor     */
  ART_ELLIPSOID = 4,  /* MirrorEllipsoidal  ModuleMirror.py:565-751  mp[0]=a, mp[1]=b                 */
  ART_CYLINDER = 5,   /* MirrorCylindrical  ModuleMirror.py:781-874  mp[0]=|R|                        */
  ART_MASK = 6,       /* Mask               ModuleMask.py:24-136                                      */
  ART_NUM_KINDS = 7
};

/* aperture kinds: ART/ModuleSupport.py `_IncludeSupport` predicates                                  */
enum ArtSupportKind {
  ART_SUP_ROUND = 0,        /* :68-70    sp = {R}                         */
  ART_SUP_ROUNDHOLE = 1,    /* :151-155  sp = {R, Rhole, cx, cy}          */
  ART_SUP_RECT = 2,         /* :228-230  sp = {X, Y}                      */
  ART_SUP_RECTHOLE = 3,     /* :322-326  sp = {X, Y, Rhole, cx, cy}       */
  ART_SUP_RECTRECTHOLE = 4  /* :431-435  sp = {X, Y, hx, hy, cx, cy}      */
};

/* element flags */
#define ART_FLAG_PERTURBED_NORMAL 1u /* IgnoreDefects=False: reflect off the defect-perturbed normal (ModuleMirror.py:933-936) */
#define ART_FLAG_ZERN_RECURRENCE 2u  /* the element's Zernike tables are in the RECURRENCE layout (any order, below)      */

/* Zernike defect table (ART/ModuleDefects.py:149-174), a DEVICE array of doubles per element.  The polynomials
 * of ART/recursive_zernike_generator.py have integer monomial coefficients; the host expands
 *     h(x,y) = sum_nm c_nm Z_nm(x,y) = sum_pq A[p][q] x^p y^q        (x, y = (P - centre) / R)
 * exactly and stores A and its two partial derivatives, so that the kernels evaluate three bivariate Horner
 * schemes instead of carrying the recurrence rows in registers (the coefficients are wave-uniform: they reach the
 * lanes through scalar loads, one copy per wave; entries of total degree > N must be zero).  Layout, for defect d:
 *   base = d * ART_ZERN_STRIDE
 *   [base+0] = R (Support._CircumCirc()),  [base+1] = max total degree N (2..ART_ZERN_MAX_ORDER)
 *   [base+2            + p*ART_ZERN_DIM + q] = A[p][q]        coefficient of x^p y^q of h
 *   [base+2 +   DIM^2  + p*ART_ZERN_DIM + q] = dA/dx [p][q]   = (p+1) A[p+1][q]
 *   [base+2 + 2*DIM^2  + p*ART_ZERN_DIM + q] = dA/dy [p][q]   = (q+1) A[p][q+1]                        */
/* Orders above ART_ZERN_MAX_ORDER (and any order, if the caller prefers): set ART_FLAG_ZERN_RECURRENCE and hand over, per
 * defect, the coefficients themselves -- [R, N, c(0,0), c(1,0), c(1,1), c(2,0), ...], c(n,m) at 2 + n(n+1)/2 + m, absent
 * terms 0, every table of one element padded to the same N <= ART_ZERN_RECURRENCE_MAX_ORDER (stride 2 + (N+1)(N+2)/2).
 * The kernels then run the reference's recurrences per ray (the monomial form is exact but loses 4e-10 of the polynomial
 * to cancellation at order 20 and everything at order 40; the recurrences stay at 1e-15).  Such elements are traced by
 * art_trace_element only (a kernel of its own, with per-lane row storage); art_trace_chain / art_scene_pack return
 * ART_ERR_UNSUPPORTED for them.                                                                                          */
#define ART_ZERN_RECURRENCE_MAX_ORDER 64
#define ART_ZERN_MAX_ORDER 16
#define ART_ZERN_DIM (ART_ZERN_MAX_ORDER + 1)                      /* 17 */
#define ART_ZERN_STRIDE (2 + 3 * ART_ZERN_DIM * ART_ZERN_DIM)      /* 869 */
#define ART_MAX_DEFECTS 16   /* tables (Zernike: one per normalisation radius) / height maps per mirror; the kernels loop */

/* Gridded height-map defect (ART/ModuleDefects.py `Fourrier` :69-146; offset lookup :131-137 through SciPy's
 * RegularGridInterpolator(method="linear")): h[ix * ny + iy] on the regular grid x = x0 + ix*dx, y = y0 + iy*dy,
 * bilinear interpolation, evaluated at (P - centre).  An array of these structs lives in DEVICE memory.
 * Points outside the grid are clamped to the edge cell (SciPy raises there; hits are inside the support, which
 * the grid covers).  Only the offset is used: the reference's Fourrier.get_normal raises under NumPy >= 1.24. */
typedef struct ArtGridDefect {
  const double* h;       /* DEVICE, nx * ny doubles */
  int32_t nx, ny;
  double x0, y0, dx, dy;
} ArtGridDefect;

/* One optical element = optic + pose.  Replaces the per-ray frame changes of
 * ART/ModuleProcessing.py:289-295, :306-309 by two constant 3x3 maps built on the host with the
 * reference's own RotationPoint special cases (ART/ModuleGeometry.py:333-343):
 *     P_optic = fwd * (P_lab - pos) + centre        u_optic = fwd * u_lab
 *     P_lab   = bwd * (P_optic - centre) + pos      u_lab   = bwd * u_optic
 * Callers fill every field except mp[2..3].  Each entry point works on its own copy of the descriptors, in which it
 * replaces bwd, pos and (torus) mp[2..3] by constants derived from the other fields (the two translations folded
 * into one offset per direction, squared radii, ...: csrc/art_device.h prepare_element()); the caller's structs are
 * never written.                                                                                       */
typedef struct ArtElementDesc {
  int32_t kind;          /* ArtOpticKind                                    */
  int32_t support_kind;  /* ArtSupportKind                                  */
  int32_t n_defects;     /* 0..ART_MAX_DEFECTS Zernike defects on a mirror  */
  uint32_t flags;        /* ART_FLAG_*                                      */
  double fwd[9];         /* row-major                                       */
  double bwd[9];         /* informational: the kernels go back through the transpose of fwd (see below) */
  double pos[3];         /* OpticalElement.position                         */
  double centre[3];      /* optic.get_centre() in the optic frame           */
  double sp[6];          /* support parameters                              */
  double mp[4];          /* mirror parameters                               */
  const double* zern;    /* DEVICE pointer, n_defects * ART_ZERN_STRIDE doubles, or NULL */
  const ArtGridDefect* grid; /* DEVICE pointer to n_grid structs, or NULL           */
  int32_t n_grid;        /* 0..ART_MAX_DEFECTS gridded height-map defects           */
  int32_t reserved;
} ArtElementDesc;

/* SoA view of one ray bundle (all DEVICE pointers, length n).  `path` is the running optical path
 * (sum of the reference's Ray.path tuple); `incidence` the incidence angle on the element the ray
 * comes from (Ray.incidence).  Ray.number / intensity / wavelength never change along a chain and are
 * therefore not part of the per-element state (slot i of every bundle is source ray i).
 * Alignment: 8 bytes are enough for correctness; for full throughput every array should start on a cache-line
 * boundary (the package pitches its rows to 512 bytes): rows that start 8 bytes off a line cost about a third of
 * the bandwidth.                                                                                             */
typedef struct ArtBundleView {
  double* ox; double* oy; double* oz;   /* Ray.point  */
  double* dx; double* dy; double* dz;   /* Ray.vector */
  double* path;
  double* incidence;
  uint8_t* alive;                       /* 1 = ray still propagating */
} ArtBundleView;

/* Plane detector (ART/ModuleDetector.py:25-62).  rot = RotationPoint(., normal, ez) as a 3x3 map
 * (ModuleDetector.py:231), built on the host like fwd/bwd.                                            */
typedef struct ArtDetectorDesc {
  double centre[3];
  double normal[3];
  double rot[9];
} ArtDetectorDesc;

int art_abi_version(void);
const char* art_last_error(void);
/* number of visible HIP devices whose architecture is gfx950 (>= 0), or a negative error code */
int art_device_count(void);

/* One element of RayTracingCalculation (ART/ModuleProcessing.py:277-311): frame change, intersection
 * (`_get_intersection` of the optic's class incl. DeformedMirror, ModuleMirror.py:969-980), support test,
 * reflection / mask transmission (ModuleMirror.py:878-939, ModuleMask.py:93-136), frame change back.
 * Reads bundle `in`, writes bundle `out` (may alias `in`).  Slots with in->alive == 0 and rays that
 * miss get out->alive = 0 and their other outputs are left untouched.                                  */
int art_trace_element(const ArtElementDesc* e, const ArtBundleView* in, const ArtBundleView* out,
                      int64_t n, void* stream);

/* Whole chain in ONE launch: the ray stays in registers from element to element.  outs[k] receives the
 * bundle after element k for every k with outs[k].alive != NULL (pass zeroed views to skip history);
 * outs[n_elems-1] is mandatory.  Same results as n_elems calls of art_trace_element for every slot that is alive in
 * the respective bundle; the other outputs of a DEAD slot are unspecified (untouched, or the ray's last live state:
 * the fused kernels store pairs of neighbouring slots with one 16-byte access).                          */
int art_trace_chain(const ArtElementDesc* elems, int32_t n_elems, const ArtBundleView* in,
                    const ArtBundleView* outs, int64_t n, void* stream);

/* The same launch with the detector read-out of the LAST bundle fused behind it (art_detector_readout's outputs and
 * statistics, ART/ModuleDetector.py:191-279): the ray is still in registers when it reaches the detector, so the
 * read-out costs its 24 B/ray of outputs instead of a second pass that re-reads 57 B/ray.  For a detector that is known
 * before the trace (manual placement, a re-trace, a scan).  Per-workgroup partial statistics go to `scratch`
 * (art_chain_readout_scratch_doubles(n) doubles, DEVICE) and are folded into out24 by a second tiny launch; the fold
 * order differs from art_detector_readout's, so sums agree with it to rounding (minima, maxima and counts exactly).
 * One launch's limit applies: n <= 2^28 (ART_ERR_UNSUPPORTED beyond; use the two separate calls).                   */
typedef struct ArtChainReadout {
  ArtDetectorDesc det;
  const double* w;       /* DEVICE weights (Ray.intensity) or NULL (w = 1)                          */
  double cx, cy, co;     /* provisional centres of the second moments, as in art_detector_readout   */
  double* X;             /* DEVICE, n doubles each, or all three NULL (statistics only)              */
  double* Y;
  double* opl;
  double* scratch;       /* DEVICE, art_chain_readout_scratch_doubles(n) doubles                    */
  double* out24;         /* DEVICE, 24 doubles: slot layout of art_detector_readout                 */
  int32_t lite;          /* 1: only count [0], sum opl [1], bounding box [2..5] and path range [12..13] are formed -- what
                            Detector.get_Delays / get_PointList2DCentre consume (ART/ModuleDetector.py:236-279); the sums
                            [6..11] and the second moments [16..21] read 0 and the weights `w` are not touched.  The tail
                            then costs 8 instead of 22 statistics per ray.  0: all 22 (default)     */
  int32_t sums;          /* 1: NO read-out -- the tail forms pass (1) of art_analyse_bundles for the last bundle instead, while the
                            ray is still in registers: out24[0..8] = count, sum point (3), sum vector (3), sum w (= count if w is
                            NULL), sum path; out24[9..23] = 0.  `det`, cx, cy, co are ignored, X / Y / opl must be NULL and lite 0.
                            For a detector that is NOT known before the trace (Detector.autoplace needs exactly these sums,
                            ART/ModuleDetector.py:109-137): hand out24 to ArtAnalysisJob.sums and the analysis of the bundle is
                            placement + ONE pass.  Same bits as the sums art_analyse_bundles forms itself (one canonical fold
                            order: tiles of 256 slots).  0 (default): a read-out                    */
} ArtChainReadout;
int64_t art_chain_readout_scratch_doubles(int64_t n);
int art_trace_chain_readout(const ArtElementDesc* elems, int32_t n_elems, const ArtBundleView* in,
                            const ArtBundleView* outs, const ArtChainReadout* readout, int64_t n, void* stream);

/* MANY chains in ONE launch (ART/ModuleProcessing.py:203-239: OEPlacement with a list-valued argument returns 10-11
 * chains that differ only in poses, which ARTmain.py:304-342 traces one after the other; the misalignment loop lists of
 * ART/ModuleOpticalChain.py:371-657 likewise).  All chains have n_elems elements and bundles of n slots.  The
 * descriptors do not fit kernel arguments, so they travel as a SCENE IMAGE in device memory:
 *   1. art_scene_bytes(n_chains, n_elems)      size of the image;
 *   2. art_scene_pack(...)                     fills a HOST buffer of that size -- pure host code, no HIP call;
 *                                              elems[c * n_elems + k], ins[c], outs[c * n_elems + k] (rules of
 *                                              art_trace_chain: outs[..].alive == NULL skips that history bundle, the
 *                                              last view of a chain and every 8th are mandatory); readouts = NULL or
 *                                              one ArtChainReadout per chain (fused read-out of each chain's last
 *                                              bundle, see art_trace_chain_readout).  Returns the scene's flags >= 0
 *                                              (bit 0: defects, bit 1: read-outs; informational -- they are also in
 *                                              the image's header), or a negative error code;
 *   3. the caller copies the image to DEVICE memory (its own allocation and memcpy);
 *   4. art_trace_scene(image_dev, image_host, n, stream)
 *                                              ONE kernel launch per 8 elements: grid.y = chain, grid.x = 256-ray tile.
 *                                              Chain count, element count and flags are read from the header of
 *                                              `image_host` (the packed host buffer image_dev is a copy of; it must stay
 *                                              valid until the call returns), never taken from the caller.
 * A launch reads nothing but the device image and the bundles: re-packing new poses into the same device buffer and
 * replaying a captured HIP graph re-traces a modified scene without host-side launch work.  Same per-ray results as
 * art_trace_chain, bit for bit.                                                                                      */
int64_t art_scene_bytes(int32_t n_chains, int32_t n_elems);
int art_scene_pack(const ArtElementDesc* elems, int32_t n_chains, int32_t n_elems, const ArtBundleView* ins,
                   const ArtBundleView* outs, const ArtChainReadout* readouts, void* image_host);
int art_trace_scene(const void* image_dev, const void* image_host, int64_t n, void* stream);

/* Bundle from array-of-structs input (the layout a caller holding ART Ray lists / (n,3) NumPy arrays has):
 * points[n][3], vectors[n][3] (DEVICE, row-major) -> SoA view; directions normalised like the Ray.vector setter
 * (ART/ModuleOpticalRay.py:85-90), path = path0[i] (or 0 if NULL), incidence = NaN, alive = 1.                      */
int art_pack_rays(const double* points, const double* vectors, const double* path0, int64_t n,
                  const ArtBundleView* out, void* stream);

/* Rigid / affine map of a whole bundle: TranslationRayList, RotationRayList, RotationAroundAxisRayList
 * (ART/ModuleGeometry.py:308-314, :372-391) -- used by the source (mis-)alignment helpers, not by tracing:
 *   point' = (rotate_points ? M * point : point) + T,   vector' = normalize(M * vector);
 * path, incidence and alive are copied.  M row-major.  `out` may alias `in`.                                        */
int art_transform_bundle(const double M[9], const double T[3], int32_t rotate_points, const ArtBundleView* in,
                         const ArtBundleView* out, int64_t n, void* stream);

/* Detector read-out (ART/ModuleDetector.py:191-234, :272-275): for every alive ray
 *   I = IntersectionLinePlane (ModuleGeometry.py:48-57);  (X,Y) = first two components of rot*(I-centre);
 *   opl = |A - I| + path.  Any of p3x..p3z / X,Y / opl may be NULL to skip that output.               */
int art_detector(const ArtDetectorDesc* d, const ArtBundleView* b, int64_t n,
                 double* p3x, double* p3y, double* p3z, double* X, double* Y, double* opl, void* stream);

/* Read-out and statistics in ONE pass over the bundle (what Detector.readout uses): the outputs of art_detector
 * plus, accumulated while the values are still in registers, the statistics of art_detector_stats in out24[0..15]
 * and second moments about provisional centres (cx, cy, co) in out24[16..23]:
 *   [16] sum (X-cx)^2  [17] sum (Y-cy)^2  [18] sum (opl-co)^2
 *   [19] sum w (X-cx)^2 [20] sum w (Y-cy)^2 [21] sum w (opl-co)^2  [22..23] 0
 * Variances follow as E[(x-c)^2] - (E[x]-c)^2; with c within a few standard deviations of the mean (cx = cy = 0 is
 * the detector centre, co an estimate of the mean path) the cancellation is harmless.  w = weights or NULL (w = 1).
 * out24: DEVICE, 24 doubles.  scratch: DEVICE, art_reduce_scratch_doubles() doubles.                              */
int art_detector_readout(const ArtDetectorDesc* d, const ArtBundleView* b, const double* w, int64_t n, double cx,
                         double cy, double co, double* p3x, double* p3y, double* p3z, double* X, double* Y,
                         double* opl, double* scratch, double* out24, void* stream);

/* Moments for a detector SCAN along its normal (autofocus, ART/ModuleProcessing.py:317-460).  For a detector shifted
 * by s along -normal (Detector.shiftByDistance(s)) every ray's read-out is exactly linear in s:
 *     X(s) = X0 + s*sx,  Y(s) = Y0 + s*sy,  opl(s) = opl0 + s*so,   so = -1/(u.normal), (sx, sy) = rot*(so*u + normal)
 * so the spot-size and duration variances at ANY s follow from global sums of per-ray products: one pass over the
 * bundle replaces one pass per scan position.  The path is centred for conditioning: O = opl0 - co, sO = so - 1.
 * out33 (DEVICE, 33 doubles), first block with weight 1, second block (+16) with weight w (= 1 if w is NULL):
 *   [0] sum 1   then for q in (X, Y, O), base = 1 + 5*k:  [base] sum q0  [base+1] sum sq  [base+2] sum q0^2
 *   [base+3] sum q0*sq  [base+4] sum sq^2
 *   [32] number of alive rays whose hit point changes side of the ray origin (t changes sign) for some shift in
 *        [0, span] (span may be negative): the reference's path |I - A| + sum(path) (ModuleDetector.py:272-275) has a
 *        kink there, so if [32] > 0 the moments describe shift 0 exactly but not the whole scan -- the caller
 *        then evaluates the scan positions one by one (a detector scanned through the last optic).            */
int art_detector_scan_moments(const ArtDetectorDesc* d, const ArtBundleView* b, const double* w, int64_t n, double co,
                              double span, double* scratch, double* out33, void* stream);

/* Masked reductions over alive rays, deterministic (fixed two-level tree, no float atomics).
 * out16 (DEVICE, 16 doubles):
 *   [0] count  [1] sum opl  [2] min X [3] max X [4] min Y [5] max Y  [6] sum X [7] sum Y
 *   [8] sum w  [9] sum w*X [10] sum w*Y [11] sum w*opl  [12] min opl [13] max opl [14..15] 0
 * w = weights (DEVICE) or NULL (then w = 1).  X, Y, opl may be NULL (their entries are then 0).
 * scratch: DEVICE, at least art_reduce_scratch_doubles() doubles.
 * Feeds get_Delays' mean (ModuleDetector.py:277), CentrePointList (ModuleGeometry.py:235-236),
 * getETransmission (ModuleAnalysisAndPlots.py:76).                                                     */
int64_t art_reduce_scratch_doubles(void);
int art_detector_stats(const uint8_t* alive, const double* X, const double* Y, const double* opl,
                       const double* w, int64_t n, double* scratch, double* out16, void* stream);

/* Second moments about given centres (two-pass variance; ModuleProcessing.py:485-532):
 * out8: [0] sum w [1] sum w (X-cx)^2 [2] sum w (Y-cy)^2 [3] sum w (opl-co)^2 [4] count [5..7] 0         */
int art_detector_moments(const uint8_t* alive, const double* X, const double* Y, const double* opl,
                         const double* w, int64_t n, double cx, double cy, double co,
                         double* scratch, double* out8, void* stream);

/* Mean ray of a bundle (FindCentralRay, ModuleProcessing.py:464-482) + sum of intensities:
 * out8: [0] count [1..3] sum point [4..6] sum vector [7] sum w                                          */
int art_bundle_sums(const ArtBundleView* b, const double* w, int64_t n, double* scratch, double* out8,
                    void* stream);

/* Gaussian intensity weights of a source bundle (ApplyGaussianIntensityToRayList, ART/ModuleSource.py:219-261):
 * w = 1 on the axis falling to `fraction` at the edge -- in angle (tan(angle)/max angle) for diverging bundles
 * (max angle to `axis` > 1e-12), in distance from the origin (|point| / max |point|) for collimated ones.
 * Two passes on the stream (max reduction, then weights); nothing returns to the host.  w_out: DEVICE, n doubles. */
int art_gaussian_intensity(const ArtBundleView* b, const double axis[3], double fraction, int64_t n,
                           double* scratch, double* w_out, void* stream);
/* The same about the bundle's OWN central ray (what ApplyGaussianIntensityToRayList passes: FindCentralRay's mean vector,
 * normalised by the Ray.vector setter, ART/ModuleProcessing.py:464-482): art_bundle_sums into `sums8` (DEVICE, 8 doubles,
 * left there for the caller), then the axis is formed ON THE DEVICE from those sums -- four launches, no host round trip
 * between the central ray and the weights (0.2 ms of waiting at the start of every OEPlacement).                          */
int art_gaussian_intensity_central(const ArtBundleView* b, double fraction, int64_t n, double* scratch, double* sums8,
                                   double* w_out, void* stream);

/* Largest angle between `axis` and the direction of any alive ray, and largest |point| (ReturnNumericalAperture,
 * ART/ModuleProcessing.py:536-566; also the first pass of art_gaussian_intensity).  out2: DEVICE, 2 doubles.          */
int art_bundle_max_angle(const ArtBundleView* b, const double axis[3], int64_t n, double* scratch, double* out2,
                         void* stream);

/* Stable compaction: idx_out[j] = slot of the j-th alive ray (source order kept, as the reference's
 * survivor lists ModuleMirror.py:928-939), *count_out = number alive.  block_counts: DEVICE scratch of
 * art_compact_scratch_ints(n) int32.                                                                    */
int64_t art_compact_scratch_ints(int64_t n);
int art_compact(const uint8_t* alive, int64_t n, int32_t* block_counts, int64_t* idx_out,
                int64_t* count_out, void* stream);

/* Deterministic sources on device (ART/ModuleSource.py:23-81, :135-169; Vogel spiral
 * ModuleGeometry.py:61-76) for global ray indices [first, first+n) of a bundle of n_total rays:
 *   kind 0: point source, half-angle `size` (rad);  kind 1: plane-wave disk of radius `size` (mm).
 * rot = RotationPoint(., ez, axis) as a 3x3 map, S = source point / disk centre.
 * Writes origin, direction, path = 0, incidence = NaN, alive = 1.                                        */
int art_make_source(int32_t kind, double size, const double rot[9], const double S[3], int64_t first,
                    int64_t n, int64_t n_total, const ArtBundleView* out, void* stream);
/* The same for the global indices first, first + step, ..., first + (n-1)*step: the STRIDED shard of rank `first` among
 * `step` ranks.  The Vogel spiral orders rays by radius, so contiguous shards of a masked or overfilled scene lose very
 * different numbers of rays; strided shards are balanced (every rank samples the whole aperture).                    */
int art_make_source_strided(int32_t kind, double size, const double rot[9], const double S[3], int64_t first,
                            int64_t step, int64_t n, int64_t n_total, const ArtBundleView* out, void* stream);

/* ExtendedSource (ART/ModuleSource.py:85-131): n_points point sources on a Vogel disk of radius `radius` (mm), each
 * emitting the same Vogel cone of rays_per_point rays with half-angle `divergence` (rad).  Global ray index
 * k = point * rays_per_point + ray-in-cone, the reference's numbering (:124-127); writes indices [first, first+n).
 * The caller derives n_points and rays_per_point from NbRays as the reference does (:111-119).               */
int art_make_extended_source(double radius, double divergence, int64_t n_points, int64_t rays_per_point,
                             const double rot[9], const double S[3], int64_t first, int64_t n,
                             const ArtBundleView* out, void* stream);

/* Multi-GPU exchange at the detector, two tiny kernels around ONE all-gather (sharding.py): every rank packs
 *   send[0..23]            its 24 read-out statistics (art_detector_readout)
 *   send[24 + 4*j + 0..3]  X, Y, opl, alive (0/1) of its sampled slot slots[j], j < k
 * all ranks all-gather their (24 + 4k)-vectors (rank-major `recv`), and art_exchange_fold folds the `world` statistics
 * vectors into the global ones (slots 2,4,12: min; 3,5,13: max; others: sum -- the slot layout of
 * art_detector_readout).  The sample parts stay where the collective put them.  All pointers DEVICE.       */
int art_exchange_pack(const double* stats24, const double* X, const double* Y, const double* opl,
                      const uint8_t* alive, const int64_t* slots, int64_t k, double* send, void* stream);
int art_exchange_fold(const double* recv, int32_t world, int64_t stride_doubles, double* stats_out24, void* stream);

/* Survivor records for the ONE gather of a sharded run (`(number:int32, X, Y, path)` = 28 B per SURVIVING
 * ray; the consumer, ART/ModuleDetector.py:254-279, sees survivors only).  Compacts the read-out of the alive slots, in
 * slot order, into `send` (DEVICE, 8-byte aligned, capacity send_bytes >= art_survivor_bytes(n, 0)):
 *   [0, 8)   int64 count                     number of alive slots
 *   [8, 16)  int64 flags                     bit 0 "dense": every slot is alive and the numbers are implicit -> the number
 *                                            section is absent (24 B per ray); bit 1 "unpacked": written by
 *                                            art_survivor_finish only (below) -- never by art_pack_survivors
 *   [16 ...) double X[count], Y[count], path[count], then int32 number[count] (absent when dense)
 * number[j] = number ? number[slot_j] : first + slot_j * step (the global index of a shard generated by
 * art_make_source / art_make_source_strided; must fit int32, ART_ERR_UNSUPPORTED otherwise).  A rank sends the first
 * art_survivor_bytes(count, dense) bytes; the header tells the receiver how to read them.  scratch_ints: DEVICE,
 * art_compact_scratch_ints(n) int32.  Everything is enqueued on `stream`; nothing returns to the host.                  */
int64_t art_survivor_bytes(int64_t count, int32_t dense);
/* ZERO-COPY form for a shard that is expected to lose nothing: let the read-out write its X, Y, opl of all n slots
 * straight into the dense layout's sections (X = send + 16, Y = X + n, path = Y + n: art_trace_chain_readout / the scene
 * read-outs / art_detector_readout take any pointers), then call art_survivor_finish with the read-out's statistics
 * (stats24[0] = number of alive slots, DEVICE): it writes the header -- (n, dense) if every slot is alive, and the buffer is
 * complete without a pack or a staging copy; otherwise (count, flags bit 1 "unpacked"): the sections hold slot-indexed
 * values with holes and the caller packs them into ANOTHER buffer with art_pack_survivors (X, Y, opl may point into this
 * one).  `send`: DEVICE, 16-byte aligned, >= art_survivor_bytes(n, 1) bytes.
 * `xhdr` (DEVICE, ART_XHDR_DOUBLES doubles, or NULL): the rank's contribution to the per-step HEADER EXCHANGE of a sharded
 * run, filled in the same launch -- [0], [1] = count, flags as int64 bit patterns, [2 .. 25] = stats24.  All ranks all-gather
 * these 208 bytes: every rank learns every shard's count (the size of the next transfer is predicted from it) and the
 * global statistics follow by art_exchange_fold(recv + 2, world, ART_XHDR_DOUBLES, out).  art_survivor_xheader builds the
 * same block behind art_pack_survivors (header read from `send`; stats24 may be NULL: zeros).                          */
#define ART_XHDR_DOUBLES 26
int art_survivor_finish(const double* stats24, int64_t n, void* send, double* xhdr, void* stream);
int art_survivor_xheader(const void* send, const double* stats24, double* xhdr, void* stream);
int art_pack_survivors(const uint8_t* alive, int64_t n, const double* X, const double* Y, const double* opl,
                       const int64_t* number, int64_t first, int64_t step, int32_t* scratch_ints, void* send,
                       int64_t send_bytes, void* stream);

/* Guide rays of the placement (ART/ModuleProcessing.py:98-126: `_singleOEPlacement` traces ONE alignment ray through the
 * chain built so far to find the direction in which the next optic is placed; OEPlacement with a list-valued argument,
 * :203-239, does that for 10-11 chains one after the other).  One launch advances the guide rays of up to `count` chains
 * by one element each: ray j meets elems[j] (any kind, the per-ray code of art_trace_element).  rays: DEVICE, count x 8
 * doubles, row j = ox, oy, oz, dx, dy, dz, path, incidence of guide ray j, updated in place; alive: DEVICE, count bytes
 * (a guide that misses its optic gets 0 and keeps its state).  The descriptors travel as kernel arguments: count <= 8 per
 * call (ART_ERR_BAD_ARG beyond; callers loop).  Elements in the Zernike recurrence layout are refused (ART_ERR_UNSUPPORTED:
 * the guide of such an optic is traced through art_trace_element).                                                       */
#define ART_GUIDES_MAX 8
int art_trace_guides(const ArtElementDesc* elems, int32_t count, double* rays, uint8_t* alive, void* stream);

/* Analysis of MANY bundles in four launches and ZERO host round trips (ART/ARTmain.py:248-300 runs, per chain of a loop
 * list: getETransmission, ART/ModuleAnalysisAndPlots.py:62-77; Detector.autoplace, ART/ModuleDetector.py:109-137;
 * FindOptimalDistance, ART/ModuleProcessing.py:369-460, or GetResultSummary, ModuleAnalysisAndPlots.py:81-129).  Per job
 * (= one bundle of n slots; all jobs of a call share n):
 *   1. sums over the alive rays: count, sum point, sum vector, sum w, sum path;
 *   2. the detector: given (mode ART_JOB_MANUAL) or placed like Detector.autoplace -- normal = -(mean vector, normalised),
 *      centre = mean point - normal * distance, refpoint = mean point; its rotation normal -> ez with the reference's
 *      special cases (ModuleGeometry.py:333-343); the provisional path centre co = mean path + the mean ray's distance
 *      to the detector (within a fraction of a millimetre of the mean optical path: second moments about it are well
 *      conditioned, so no pass is spent on finding the exact mean);
 *   3. ONE pass of read-out moments on that detector: the 32 sums of art_detector_scan_moments (the read-out of every ray
 *      is linear in a shift s of the detector along its normal, so spot size and duration at ANY s follow from them --
 *      a whole autofocus search without touching the bundle again), the bounding box of (X, Y) and the range of the
 *      optical path at s = 0, the largest angle between a ray and the mean vector (ReturnNumericalAperture,
 *      ModuleProcessing.py:536-566), and the two shifts nearest to 0 at which some ray's hit point passes through its
 *      origin (there |I - A| has a kink and the linear model ends: callers evaluate scans beyond them position by position).
 * Jobs with mode ART_JOB_SUMS stop after step 1 (e.g. the source bundle, for the transmission's denominator).
 * jobs_dev: DEVICE copy of the HOST array jobs_host (the caller's memcpy; jobs_host is read for validation only and must
 * hold the same contents).  out: DEVICE, n_jobs x ART_ANALYSIS_DOUBLES doubles, job-major:
 *   [0] count  [1..3] sum point  [4..6] sum vector  [7] sum w (= count if w is NULL)  [8] sum path  [9] 0
 *   [10..12] detector centre  [13..15] detector normal  [16..18] refpoint (mean point; manual: job.refpoint)  [19] co
 *   [20..51] the 32 moment sums of art_detector_scan_moments (path centred on co)  [52] 0
 *   [53] largest shift s <= 0 and [54] smallest shift s > 0 at which a ray's t changes sign (-inf / +inf if none)
 *   [55] largest angle to the mean vector (rad)
 *   [56] min X [57] max X [58] min Y [59] max Y [60] min opl [61] max opl   (s = 0; +inf / -inf if nothing is alive)
 *   [62..63] 0
 * One call covers bundles of n <= 2^28 slots (ART_ERR_UNSUPPORTED beyond: 32-bit buffer offsets, as in the tracing launches).
 * A job without alive rays gets count 0, NaN detector fields and zero moments.  A manual detector's normal must be a unit
 * vector (| |normal|^2 - 1 | <= 1e-12, ART_ERR_BAD_ARG otherwise); a manual detector parallel to the mean ray (mean vector .
 * normal = 0) makes the provisional path centre infinite and every path moment NaN -- as the reference's read-out of such a
 * detector is.  scratch: DEVICE, art_analysis_scratch_doubles(n_jobs, n) doubles.  Deterministic (fixed grids, fixed fold order, no float atomics); the
 * launches are enqueued on `stream`, nothing returns to the host.                                                        */
#define ART_ANALYSIS_DOUBLES 64
enum ArtJobMode { ART_JOB_AUTOPLACE = 0, ART_JOB_MANUAL = 1, ART_JOB_SUMS = 2 };
typedef struct ArtAnalysisJob {
  ArtBundleView b;        /* the analysed bundle                                                       */
  const double* w;        /* DEVICE weights (Ray.intensity) or NULL (w = 1)                            */
  double distance;        /* ART_JOB_AUTOPLACE: DistanceDetector                                       */
  int32_t mode;           /* ArtJobMode                                                                */
  int32_t reserved;
  double centre[3];       /* ART_JOB_MANUAL: the detector (unit normal, used bit for bit)              */
  double normal[3];
  double refpoint[3];
  const double* sums;     /* DEVICE, 9 doubles, or NULL: the sums of step 1 if something formed them already (the tail of
                             the tracing launch, ArtChainReadout.sums -> its out24).  Jobs that bring them skip step 1; if all
                             do, the bundles are read ONCE (step 3).                                   */
} ArtAnalysisJob;
int64_t art_analysis_scratch_doubles(int32_t n_jobs, int64_t n);
int art_analyse_bundles(const ArtAnalysisJob* jobs_dev, const ArtAnalysisJob* jobs_host, int32_t n_jobs, int64_t n,
                        double* scratch, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ART_HIP_H */
