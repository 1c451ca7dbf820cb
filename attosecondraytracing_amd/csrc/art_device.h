// art_device.h -- per-ray math of the ART hot path, one ray per lane.
//
// Written for gfx950 (wave64, fp64 VALU); the same functions compile with g++ (-DART_HOST_TWIN) into the
// CPU twin that tests/ use to check the kernel math in a container without a GPU (oracle/twin/).
// Every function cites the reference lines whose result it reproduces (paths relative to /root/reference).
//
// Differences from the reference that are deliberate (all far inside the 1e-10 parity tolerance):
//   * the four per-ray frame rotations of ModuleProcessing.py:289-295/:306-309 are two constant 3x3 maps;
//   * quadrics are solved in closed form (cancellation-free) instead of np.roots;
//   * the torus is NOT solved through its expanded quartic (coefficients ~1e15): candidates with
//     z < -R lie on the outer half-tube, which is part of the boundary of the convex body
//     K = disk(R) (+) ball(r); a line meets it at most twice, and H(t) = dist(P(t), disk)^2 - r^2 is
//     convex in t, so monotone Newton from outside K (start: an osculating spheroid that contains K) finds exactly
//     the reference's candidates; self-intersecting tori add the inner convex body of the quartic's second factor.
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/art_hip.h"

#if defined(__HIPCC__) && !defined(ART_HOST_TWIN)
#define ART_HD __host__ __device__ __forceinline__
#define ART_DEVICE_CODE 1
#else
#define ART_HD inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// wave64 ballot: keep the whole wavefront in the Newton loop until every lane has converged
#define ART_WAVE_ANY(pred) (__ballot(pred) != 0ull)
#else
#define ART_WAVE_ANY(pred) (pred)
#endif

namespace art {

struct Ray {
  double ox, oy, oz, dx, dy, dz, path, inc;
};

// slots of a PREPARED descriptor (prepare_element() below says what they hold)
#define ART_D_IN_OFF bwd
#define ART_D_OUT_OFF bwd + 3
#define ART_D_R2 bwd[6]
#define ART_D_BOX_RHO2 bwd[7]
#define ART_D_BOX_Y2 bwd[8]
#define ART_D_RB pos[0]
#define ART_D_SUM2 pos[1]
#define ART_D_DIF2 pos[2]
#define ART_D_I2R mp[1]              /* torus: 1/(2R) in place of r (r^2 is ART_D_R2) */
#define ART_FLAG_D_LEMON 0x80000000u /* torus with r > R: the quartic's second factor has real roots too */

// ---------------------------------------------------------------------------------------------------------
// small helpers
ART_HD double dot3(double ax, double ay, double az, double bx, double by, double bz) {
  return fma(ax, bx, fma(ay, by, az * bz));
}

ART_HD void mat3_apply(const double* M, double x, double y, double z, double& rx, double& ry, double& rz) {
  rx = fma(M[0], x, fma(M[1], y, M[2] * z));
  ry = fma(M[3], x, fma(M[4], y, M[5] * z));
  rz = fma(M[6], x, fma(M[7], y, M[8] * z));
}

// M p + o and M^T p + o.  The offset is added LAST: M and o are both wave-uniform (scalar registers), an instruction
// takes one scalar operand, and as the innermost addend of the FMA chain o would first be copied to vector registers.
ART_HD void mat3_apply_off(const double* M, const double* o, double x, double y, double z, double& rx, double& ry, double& rz) {
  rx = fma(M[0], x, fma(M[1], y, M[2] * z)) + o[0];
  ry = fma(M[3], x, fma(M[4], y, M[5] * z)) + o[1];
  rz = fma(M[6], x, fma(M[7], y, M[8] * z)) + o[2];
}
ART_HD void mat3t_apply_off(const double* M, const double* o, double x, double y, double z, double& rx, double& ry, double& rz) {
  rx = fma(M[0], x, fma(M[3], y, M[6] * z)) + o[0];
  ry = fma(M[1], x, fma(M[4], y, M[7] * z)) + o[1];
  rz = fma(M[2], x, fma(M[5], y, M[8] * z)) + o[2];
}

ART_HD void mat3t_apply(const double* M, double x, double y, double z, double& rx, double& ry, double& rz) {
  // transpose of M: the frame maps are orthogonal (rotations and the point inversion -I), so the way back
  // is the transpose of the way in
  rx = fma(M[0], x, fma(M[3], y, M[6] * z));
  ry = fma(M[1], x, fma(M[4], y, M[7] * z));
  rz = fma(M[2], x, fma(M[5], y, M[8] * z));
}

// ---------------------------------------------------------------------------------------------------------
// fp64 reciprocal / reciprocal square root built on the gfx950 hardware seeds (v_rcp_f64 / v_rsq_f64, ~2^-23
// relative) and fused Newton-Raphson steps.  Operands on this path are ordinary magnitudes (mm-scale lengths,
// unit-vector dot products): no denormal/overflow scaling is needed, which is what makes these sequences a
// third of the instruction count of the IEEE division / square root expansions.
ART_HD double rcp_seed(double x) {  // ~1e-7 relative: enough for a Newton STEP (the iteration self-corrects)
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcp(x);
#else
  return 1.0 / x;
#endif
}
ART_HD double rcp_full(double x) {  // <= 1 ulp-level: two Newton-Raphson steps on the seed
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
#else
  return 1.0 / x;
#endif
}
ART_HD double div_full(double a, double b) {  // a / b with one residual correction
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = rcp_full(b);
  const double q = a * y;
  return fma(fma(-b, q, a), y, q);
#else
  return a / b;
#endif
}
ART_HD double rsqrt_full(double x) {  // 1/sqrt(x), x > 0 finite
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}
// sqrt(x) and 1/sqrt(x) together; sqrt gets a final residual correction (error < 1 ulp for x > 0)
ART_HD void sqrt_rsqrt(double x, double& s, double& rs) {
#if defined(__HIP_DEVICE_COMPILE__)
  rs = rsqrt_full(x);
  const double g = x * rs;
  s = fma(fma(-g, g, x), 0.5 * rs, g);
#else
  s = sqrt(x);
  rs = 1.0 / s;
#endif
}
// sqrt(x) to < 1 ulp together with a 1/sqrt(x) that is only good to ~1e-13 relative (one coupled Goldschmidt step on
// the seed, then a residual correction of the root): for the torus function, where 1/rho scales a derivative.
ART_HD void sqrt_rsqrt_coarse(double x, double& s, double& rs) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(x);
  const double g = x * y, h = 0.5 * y;
  const double e = fma(-g, h, 0.5);
  const double g1 = fma(g, e, g), h1 = fma(h, e, h);
  s = fma(fma(-g1, g1, x), h1, g1);
  rs = h1 + h1;
#else
  s = sqrt(x);
  rs = 1.0 / s;
#endif
}
ART_HD double sqrt_seed(double x) {  // ~1e-7 relative: starting points only
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrt(x);  // the bare v_sqrt_f64 (sqrtf() on a converted operand expands to 17 instructions)
#else
  return sqrt(x);
#endif
}

// a * b + c with c a compile-time constant (or any wave-uniform value): the VOP3 form with c as its one scalar
// operand.  v_fma_f64 cannot take a 64-bit literal, and left to itself the compiler selects the two-address
// v_fmac_f64, whose addend must sit in a VGPR -- two v_mov_b32 per constant, on the vector pipe that bounds the tracing
// kernels.  This way the constant is built by two s_mov_b32 on the scalar pipe, which has slots to spare.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double fma_sc(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
  return d;
}
#else
ART_HD double fma_sc(double a, double b, double c) { return fma(a, b, c); }
#endif

// atan(q) for 0 <= q <= 1 (half the Kahan angle): table-free argument reduction to |t| <= tan(pi/16) by
// atan(q) = atan(c) + atan((q - c) / (1 + q c)), c in {0, tan(pi/8), tan(pi/4)... } then an odd polynomial.
ART_HD double atan01(double q) {
  // breakpoints c_k = tan(k pi/8), k = 0..2; intervals split at tan(pi/16), tan(3pi/16)
  double c = 0.0, ac = 0.0;
  if (q > 0.19891236737965800691) { c = 0.41421356237309504880; ac = 0.39269908169872415481; }
  if (q > 0.66817863791929891999) { c = 1.0; ac = 0.78539816339744830962; }
  // (q - c) / (1 + q c) through the refined reciprocal: 1.5 ulp, no residual correction needed for an angle
  const double t = (q - c) * rcp_full(fma(q, c, 1.0));
  const double z = t * t;  // |t| <= tan(pi/16) = 0.1989: z <= 0.0396, series to z^10 is < 1e-16 relative
  double p = -1.0 / 21.0;
  p = fma_sc(p, z, 1.0 / 19.0);
  p = fma_sc(p, z, -1.0 / 17.0);
  p = fma_sc(p, z, 1.0 / 15.0);
  p = fma_sc(p, z, -1.0 / 13.0);
  p = fma_sc(p, z, 1.0 / 11.0);
  p = fma_sc(p, z, -1.0 / 9.0);
  p = fma_sc(p, z, 1.0 / 7.0);
  p = fma_sc(p, z, -1.0 / 5.0);
  p = fma_sc(p, z, 1.0 / 3.0);
  return ac + fma(-t * z, p, t);
}

// Kahan angle between two UNIT vectors, ART/ModuleGeometry.py:40-44: 2*atan2(|U-V|, |U+V|) (the reference scales by
// the two norms first; they are 1 +- 1e-16 here).  `duv` = U.V, which the caller has anyway.  Of the two norms only
// the SMALLER is formed from its vector -- that is the one Kahan's formula protects from cancellation; for unit
// vectors |U-V|^2 + |U+V|^2 = 4, so the larger (>= 2) follows by a subtraction at full relative accuracy.
ART_HD double kahan_angle_unit(double ux, double uy, double uz, double vx, double vy, double vz, double duv) {
  const bool acute = duv >= 0.0;                       // |U-V| <= |U+V|
  const double sg = -copysign(1.0, duv);
  const double wx = fma(sg, vx, ux), wy = fma(sg, vy, uy), wz = fma(sg, vz, uz);
  const double lo = dot3(wx, wy, wz, wx, wy, wz), hi = 4.0 - lo;
  // tan(angle/2) = sqrt(lo/hi) = lo / sqrt(lo*hi); lo = 0 (U = +-V exactly) and denormal lo read as angle 0
  const double q = lo * rsqrt_full(fmax(lo, 1e-290) * hi);
  const double h = atan01(q);
  return acute ? 2.0 * h : fma(-2.0, h, 3.14159265358979323846);
}

// ART/ModuleGeometry.py:249-268, ART/ModuleSupport.py:68-70,:151-155,:228-230,:322-326,:431-435
ART_HD bool in_disk(double R, double x, double y) { return fma(x, x, y * y) <= R * R; }
ART_HD bool in_rect(double X, double Y, double x, double y) {
  return fabs(x) <= fabs(X * 0.5) && fabs(y) <= fabs(Y * 0.5);
}
ART_HD bool include_support(int kind, const double* sp, double x, double y) {
  switch (kind) {
    case ART_SUP_ROUND: return in_disk(sp[0], x, y);
    case ART_SUP_ROUNDHOLE: return in_disk(sp[0], x, y) && !in_disk(sp[1], x - sp[2], y - sp[3]);
    case ART_SUP_RECT: return in_rect(sp[0], sp[1], x, y);
    case ART_SUP_RECTHOLE: return in_rect(sp[0], sp[1], x, y) && !in_disk(sp[2], x - sp[3], y - sp[4]);
    default: return in_rect(sp[0], sp[1], x, y) && !in_rect(sp[2], sp[3], x - sp[4], y - sp[5]);
  }
}

// Real roots of a t^2 + b t + c (SolverQuadratic, ART/ModuleGeometry.py:80-91).  np.roots drops an exactly
// zero leading coefficient (-> one root -c/b); for tiny a it returns the accurate small root plus a huge
// one -- reproduced by the cancellation-free form.  Returns the number of real roots (0, 1 or 2).
ART_HD int quadratic_roots(double a, double b, double c, double& t1, double& t2) {
  if (a == 0.0) {
    if (b == 0.0) return 0;
    t1 = -c / b;
    t2 = t1;
    return 1;
  }
  double disc = fma(b, b, -4.0 * a * c);
  if (!(disc >= 0.0)) return 0;
  double sq = 0.0, rs;
  if (disc > 0.0) sqrt_rsqrt(disc, sq, rs);
  double q = -0.5 * (b + copysign(sq, b));
  if (q == 0.0) {  // b == 0 and c == 0: double root at 0
    t1 = 0.0;
    t2 = 0.0;
    return 2;
  }
  t1 = q / a;          // a may be tiny (1e-34 for axis-parallel rays on a parabola): keep the IEEE division
  t2 = div_full(c, q);
  return 2;
}

// ---------------------------------------------------------------------------------------------------------
// Zernike defects (ART/ModuleDefects.py:149-174; polynomials of ART/recursive_zernike_generator.py:35-254).
// The host expands the summed surface into monomials (exact: the recurrences have integer coefficients) and hands
// over the polynomial and its two partial derivatives as dense [p][q] tables (layout: art_hip.h).  Per ray this is a
// bivariate Horner scheme: row p (the coefficient of x^p) is a polynomial in y, the rows are folded by a Horner
// scheme in x.
//
// Coefficients are the same for every ray, i.e. WAVE-UNIFORM.  The evaluators are specialised at compile time per
// order bucket (NMAX = 4, 8, 12, 16; a table of order N runs in the smallest bucket >= N, the entries of degree > N
// are zero and a leading zero leaves a Horner chain unchanged bit for bit), fully unrolled, and read the table
// through a pointer type chosen by the caller:
//   * `zuni_t` (device default): the constant address space -> scalar loads (s_load_dwordx2..x16 through the scalar
//     cache) into SGPRs, one copy per WAVE; v_fma_f64 takes the coefficient as its scalar operand.  No LDS, no
//     staging pass, no barrier; the unrolled rows are independent dependency chains the scheduler interleaves.
//   * a plain pointer into LDS (-DART_ZERN_LDS, kept for the comparison recorded in DESIGN.md): the dense tables are
//     staged once per workgroup and every lane reads the same LDS address (broadcast) -- 64 copies per wave of every
//     coefficient through the LDS pipe, which is what bounded the order-16 surface in round 1.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ART_ZERN_LDS)
typedef const double __attribute__((address_space(4)))* zuni_t;
#define ART_ZUNI(p) ((art::zuni_t)(p))
// a * b + c with c wave-uniform: the VOP3 form with the coefficient as its one scalar operand.  Left to itself the
// compiler selects the two-address v_fmac_f64 (d += a * b), whose addend must sit in a VGPR, and copies every
// coefficient there first -- two v_mov_b32 per FMA, i.e. three times the VALU work, and 288 VGPRs of live copies.
__device__ __forceinline__ double fma_uc(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
  return d;
}
#else
typedef const double* zuni_t;
#define ART_ZUNI(p) (p)
ART_HD double fma_uc(double a, double b, double c) { return fma(a, b, c); }
#endif

#define ART_ZPOLY (ART_ZERN_DIM * ART_ZERN_DIM)   /* doubles per dense polynomial table */

ART_HD constexpr int zern_bucket(int N) { return N <= 4 ? 4 : (N <= 8 ? 8 : (N <= 12 ? 12 : 16)); }

// h = get_offset (:168-174)
template <int NMAX, typename TP>
ART_HD double zernike_offset_t(TP tab, double px, double py) {
  const double iR = rcp_full(tab[0]);
  const double x = px * iR, y = py * iR;
  double acc = 0.0;
#pragma unroll
  for (int p = NMAX; p >= 0; --p) {
    double in = tab[2 + p * ART_ZERN_DIM + NMAX - p];
#pragma unroll
    for (int q = NMAX - p - 1; q >= 0; --q) in = fma_uc(in, y, tab[2 + p * ART_ZERN_DIM + q]);
    acc = fma(acc, x, in);
  }
  return acc;
}

// (gX, gY): get_normal (:159-166) returns (-gX, -gY, 1); both derivative polynomials (degree N-1) in one pass
template <int NMAX, typename TP>
ART_HD void zernike_slopes_t(TP tab, double px, double py, double& gX, double& gY) {
  const double iR = rcp_full(tab[0]);
  const double x = px * iR, y = py * iR;
  double ax = 0.0, ay = 0.0;
#pragma unroll
  for (int p = NMAX - 1; p >= 0; --p) {
    double ix = tab[2 + ART_ZPOLY + p * ART_ZERN_DIM + NMAX - 1 - p];
    double iy = tab[2 + 2 * ART_ZPOLY + p * ART_ZERN_DIM + NMAX - 1 - p];
#pragma unroll
    for (int q = NMAX - 2 - p; q >= 0; --q) {
      ix = fma_uc(ix, y, tab[2 + ART_ZPOLY + p * ART_ZERN_DIM + q]);
      iy = fma_uc(iy, y, tab[2 + 2 * ART_ZPOLY + p * ART_ZERN_DIM + q]);
    }
    ax = fma(ax, x, ix);
    ay = fma(ay, x, iy);
  }
  gX = ax * iR;
  gY = ay * iR;
}

// bucket dispatch on the table's order (wave-uniform branch)
template <typename TP>
ART_HD double zernike_offset(TP tab, double px, double py) {
  const int N = (int)tab[1];
  if (N <= 4) return zernike_offset_t<4>(tab, px, py);
  if (N <= 8) return zernike_offset_t<8>(tab, px, py);
  if (N <= 12) return zernike_offset_t<12>(tab, px, py);
  return zernike_offset_t<16>(tab, px, py);
}
template <typename TP>
ART_HD void zernike_slopes(TP tab, double px, double py, double& gX, double& gY) {
  const int N = (int)tab[1];
  if (N <= 4) return zernike_slopes_t<4>(tab, px, py, gX, gY);
  if (N <= 8) return zernike_slopes_t<8>(tab, px, py, gX, gY);
  if (N <= 12) return zernike_slopes_t<12>(tab, px, py, gX, gY);
  return zernike_slopes_t<16>(tab, px, py, gX, gY);
}

// ---------------------------------------------------------------------------------------------------------
// Zernike surfaces of ANY order (ART_FLAG_ZERN_RECURRENCE; orders above ART_ZERN_MAX_ORDER): the Cartesian recurrences of
// ART/recursive_zernike_generator.py:51-246 themselves, per ray.  The monomial expansion the Horner evaluators above use
// is exact but ill-conditioned (integer coefficients reach 1e8 at order 20 and 5e17 at order 40, i.e. 4e-10 and 1e-2 of
// cancellation error relative to the polynomial), the recurrences are stable (3e-15 at order 40).  Row n needs rows
// n - 1 and n - 2 of the values and, for the derivatives, row n - 2 of each derivative: three rotating rows of
// Z, dZ/dx, dZ/dy live in per-lane arrays (private memory on the GPU: this path has its own kernel, so that the
// register-resident kernels stay free of scratch).  Table layout (include/art_hip.h): [R, N, c(0,0), c(1,0), c(1,1),
// c(2,0), ...] with c(n, m) at 2 + n (n + 1) / 2 + m, absent terms 0.
#define ART_ZGEN_MAX_ORDER 64
ART_HD constexpr int zgen_stride(int N) { return 2 + (N + 1) * (N + 2) / 2; }

ART_HD void zernike_recurrence(const double* tab, double px, double py, bool want_grad, double& h, double& gX, double& gY) {
  const int N = (int)tab[1];
  const double iR = 1.0 / tab[0];
  const double x = px * iR, y = py * iR;
  const double* c = tab + 2;
  double Z[3][ART_ZGEN_MAX_ORDER + 1], GX[3][ART_ZGEN_MAX_ORDER + 1], GY[3][ART_ZGEN_MAX_ORDER + 1];
  // rows 0 and 1 (:51-62)
  Z[0][0] = 1.0; GX[0][0] = 0.0; GY[0][0] = 0.0;
  Z[1][0] = y; GX[1][0] = 0.0; GY[1][0] = 1.0;
  Z[1][1] = x; GX[1][1] = 1.0; GY[1][1] = 0.0;
  double s = c[0] + c[1] * y + c[2] * x, sx = c[2], sy = c[1];
  for (int n = 2; n <= N; ++n) {
    const int a = n % 3, b = (n + 2) % 3, d = (n + 1) % 3;      // rows n, n - 1, n - 2
    const double* Zb = Z[b];
    const double* Zd = Z[d];
    const double fn = (double)n;
    const double* cn = c + n * (n + 1) / 2;
    for (int m = 0; m <= n; ++m) {
      double z, gx, gy;
      if (m == 0) {                                  // :79-95
        z = x * Zb[0] + y * Zb[n - 1];
        gx = fn * Zb[0];
        gy = fn * Zb[n - 1];
      } else if (m == n) {                           // :97-110
        z = x * Zb[n - 1] - y * Zb[0];
        gx = fn * Zb[n - 1];
        gy = -1.0 * fn * Zb[0];
      } else if ((n & 1) && 2 * m == n - 1) {        // :112-145
        z = y * Zb[n - 1 - m] + x * Zb[m - 1] - y * Zb[n - m] - Zd[m - 1];
        gx = fn * Zb[m - 1] + GX[d][m - 1];
        gy = fn * Zb[n - 1 - m] - fn * Zb[n - m] + GY[d][m - 1];
      } else if ((n & 1) && 2 * m == n + 1) {        // :147-177
        z = x * Zb[m] + y * Zb[n - 1 - m] + x * Zb[m - 1] - Zd[m - 1];
        gx = fn * Zb[m] + fn * Zb[m - 1] + GX[d][m - 1];
        gy = fn * Zb[n - 1 - m] + GY[d][m - 1];
      } else if (!(n & 1) && 2 * m == n) {           // :179-209
        z = 2.0 * x * Zb[m] + 2.0 * y * Zb[m - 1] - Zd[m - 1];
        gx = 2.0 * fn * Zb[m] + GX[d][m - 1];
        gy = 2.0 * fn * Zb[n - 1 - m] + GY[d][m - 1];
      } else {                                       // :211-246
        z = x * Zb[m] + y * Zb[n - 1 - m] + x * Zb[m - 1] - y * Zb[n - m] - Zd[m - 1];
        gx = fn * Zb[m] + fn * Zb[m - 1] + GX[d][m - 1];
        gy = fn * Zb[n - 1 - m] - fn * Zb[n - m] + GY[d][m - 1];
      }
      Z[a][m] = z; GX[a][m] = gx; GY[a][m] = gy;
      const double cm = cn[m];
      s = fma(cm, z, s);
      if (want_grad) { sx = fma(cm, gx, sx); sy = fma(cm, gy, sy); }
    }
  }
  h = s;
  gX = sx * iR;
  gY = sy * iR;
}

// Gridded height map: bilinear lookup (ART/ModuleDefects.py:131-137; SciPy RegularGridInterpolator, linear)
ART_HD double grid_offset(const ArtGridDefect& g, double px, double py) {
  // Cell coordinates clamped to the grid BEFORE the integer conversion (a NaN or huge coordinate must not reach the
  // cast; fmin/fmax drop a NaN): a point outside the map reads the edge value.  Hits lie inside the mirror's support,
  // which the map covers -- the host shell refuses a map that does not (the reference's interpolator raises for such
  // a point, ART/ModuleDefects.py:108-110).
  const double fx = fmin(fmax((px - g.x0) / g.dx, 0.0), (double)(g.nx - 1));
  const double fy = fmin(fmax((py - g.y0) / g.dy, 0.0), (double)(g.ny - 1));
  int ix = (int)fx, iy = (int)fy;
  ix = ix > g.nx - 2 ? g.nx - 2 : ix;
  iy = iy > g.ny - 2 ? g.ny - 2 : iy;
  const double tx = fx - (double)ix, ty = fy - (double)iy;
  const double* r0 = g.h + (int64_t)ix * g.ny + iy;
  const double* r1 = r0 + g.ny;
  const double v00 = r0[0], v01 = r0[1], v10 = r1[0], v11 = r1[1];
  return (v00 * (1.0 - tx) + v10 * tx) * (1.0 - ty) + (v01 * (1.0 - tx) + v11 * tx) * ty;
}

// ---------------------------------------------------------------------------------------------------------
// undeformed normals, get_normal of each mirror class
template <int KIND>
ART_HD void base_normal(const ArtElementDesc& e, double x, double y, double z, double& nx, double& ny, double& nz) {
  if (KIND == ART_PLANE || KIND == ART_MASK) {  // ModuleMirror.py:84-87, ModuleMask.py:63-66
    nx = 0.0; ny = 0.0; nz = 1.0;
    return;
  }
  double gx, gy, gz;
  if (KIND == ART_SPHERE) {  // :180-183  normalize(-P)
    gx = -x; gy = -y; gz = -z;
  } else if (KIND == ART_PARABOLA) {  // :349-355  normalize(-x, -y, p)
    gx = -x; gy = -y; gz = e.mp[0];
  } else if (KIND == ART_TORUS) {  // :480-498  -grad of the quartic form, 4x(S + A) - 8xR^2 = 4x(S - R^2 - r^2)
    const double S = dot3(x, y, z, x, y, z);
    const double kxz = S - e.ART_D_SUM2, ky = S + e.ART_D_DIF2;   // R^2 + r^2, R^2 - r^2
    gx = -x * kxz; gy = -y * ky; gz = -z * kxz;
  } else if (KIND == ART_ELLIPSOID) {  // :685-693
    // same direction as (-x/a^2, -y/b^2, -z/b^2), scaled by a^2 b^2 to avoid two divisions
    const double a2 = e.mp[0] * e.mp[0], b2 = e.mp[1] * e.mp[1];
    gx = -x * b2; gy = -y * a2; gz = -z * a2;
  } else {  // cylinder :846-849
    gx = 0.0; gy = -y; gz = -z;
  }
  const double inv = rsqrt_full(dot3(gx, gy, gz, gx, gy, gz));
  nx = gx * inv; ny = gy * inv; nz = gz * inv;
}

// ---------------------------------------------------------------------------------------------------------
// torus.  The reference's quartic (ModuleMirror.py:450-465) is the product of two factors,
//   [(rho - R)^2 + y^2 - r^2] * [(rho + R)^2 + y^2 - r^2] = 0,   rho = sqrt(x^2 + z^2):
// the torus proper and, only when r > R, a spurious inner "lemon" rho = sqrt(r^2 - y^2) - R that np.roots
// also returns (and that the reference therefore reflects off).  Both are handled through convex functions
// of the ray parameter:
//   SIDE = -1:  H(t) = dist(P, disk(R))^2 - r^2          (body K = disk(R) (+) ball(r); outer half-tube)
//   SIDE = +1:  G(t) = maxdist(P, circle(R))^2 - r^2     (body L = intersection of balls B(c, r), c on the circle)
template <int SIDE>
ART_HD void torus_F(double R, double r2, double x, double y, double z, double ux, double uy, double uz,
                    double& F, double& dF) {
  // branch-free: on the torus axis (rho = 0) the clamp keeps 1/rho finite and the side selection below ignores it
  const double rho2 = fmax(fma(x, x, z * z), 1e-300);
  double rho, irho;
  sqrt_rsqrt_coarse(rho2, rho, irho);
  const double drho = fma(x, ux, z * uz) * irho;  // d rho / dt
  // SIDE < 0: distance to the disk is |y| above/below it (rho <= R) -> a = max(rho - R, 0);  SIDE > 0: a = rho + R
  const double a = (SIDE < 0) ? fmax(rho - R, 0.0) : rho + R;
  F = fma(a, a, fma(y, y, -r2));
  dF = 2.0 * fma(a, drho, y * uy);
}

// Monotone Newton on the convex F from `t` towards the root on the side given by `dir`
// (+1: start right of the exit root, move left; -1: start left of the entry root, move right).
// Returns true and the root in t, or false when the line misses the body.  The body is written with selects, not
// branches: all lanes of a wavefront run the same few iterations and leave together through the ballot.
template <int SIDE>
ART_HD bool torus_newton(double R, double r2, double Ax, double Ay, double Az, double ux, double uy, double uz,
                         double dir, bool want, double& t) {
  bool active = want, found = false;
  int it = 0;
  while (ART_WAVE_ANY(active)) {
    double F, dF;
    torus_F<SIDE>(R, r2, fma(t, ux, Ax), fma(t, uy, Ay), fma(t, uz, Az), ux, uy, uz, F, dF);
    // slope of the wrong sign: past the minimum of a convex function without having met a root on this side
    const bool miss = !(dF * dir > 0.0);
    // the step may use the 1e-7-accurate hardware reciprocal: Newton corrects itself, and the LAST step
    // (|dt| <= 1e-7 |t|) then carries an error <= 1e-7 |dt| ~ 1e-14 |t|
    const double dt = F * rcp_seed(dF);
    const double tn = t - dt;
    ++it;
    // quadratic convergence: the step just taken leaves an error ~ dt^2 * F''/(2F') ~ 1e-14 (1+|t|)^2/(r cos)
#ifdef ART_DIAG_NEWTON1
    const bool conv = true;
#else
    const bool conv = (fabs(dt) <= 1e-7 * (1.0 + fabs(tn))) || (it >= 60);
#endif
    const bool step = active && !miss;
    t = step ? tn : t;
    found = found || (step && conv);
    active = step && !conv;
  }
  return found;
}

// Positive-side roots (entry, exit) of the line with one of the two convex bodies; `rb` = radius of a sphere
// about the origin that contains the body (R + r for K, sqrt(r^2 - R^2) for L).  Returns a mask: bit 0 = entry root
// in ta, bit 1 = exit root in tb.
// SIDE = -1 uses a tighter start than the bounding sphere: the oblate spheroid  rho^2/(R+r)^2 + y^2/(r(R+r)) = 1
// contains K [squaring  (R+r) sqrt(1 - y^2/(r(R+r))) >= R + sqrt(r^2 - y^2)  leaves (R/r)(r - sqrt(r^2-y^2))^2 >= 0]
// and osculates the outer half-tube along the equator: the gap is of 4th order in y, so for a mirror (rays a few mm
// to cm off the equatorial plane, r ~ 100s of mm) Newton starts within ~1e-2 mm of the root and needs 2 steps where
// the bounding sphere needed 4.  `ia2` = 1/(R+r)^2, `ic2` = 1/(r(R+r)) (prepare_element()).
template <int SIDE>
ART_HD int torus_body_roots(double R, double r2, double rb, double ia2, double ic2, double box_rho2, double box_y2,
                            double dif2, double i2R, double Ax, double Ay, double Az, double ux, double uy, double uz,
                            double& ta, double& tb) {
  // bounding quadric  qa t^2 + 2 hb t + c = 0.  Only starting points are needed, so a single-precision sqrt is
  // enough; the miss test keeps a safety margin for it.
  double qa, hb, c, iqa;
  if (SIDE < 0) {
    qa = fma(fma(ux, ux, uz * uz), ia2, uy * uy * ic2);
    hb = fma(fma(Ax, ux, Az * uz), ia2, Ay * uy * ic2);
    c = fma(fma(Ax, Ax, Az * Az), ia2, fma(Ay * Ay, ic2, -1.0));
    iqa = rcp_seed(qa);
    iqa = fma(fma(-qa, iqa, 1.0), iqa, iqa);
  } else {
    qa = 1.0; iqa = 1.0;   // |u| = 1
    hb = dot3(Ax, Ay, Az, ux, uy, uz);
    c = dot3(Ax, Ay, Az, Ax, Ay, Az) - rb * rb;
  }
  const double disc = fma(hb, hb, -qa * c);
  bool any = disc >= -1e-6 * fma(hb, hb, qa * fabs(c) + qa);
  const double sq = (disc > 0.0) ? sqrt_seed(disc) : 0.0;
  const double tm = -hb * iqa;                  // closest approach to the centre
  const double pad = 1e-6 * (rb + fabs(tm));
  const double ts1 = fma(-sq, iqa, tm) - pad, ts2 = fma(sq, iqa, tm) + pad;
  any = any && (ts2 > 1e-12);
  // Is the ray origin inside the body?  Usual case for a mirror (the previous optic sits inside the tube): decided
  // without a square root by the inscribed box |y| < 0.7 r, rho < R + 0.7 r (0.7^2 + 0.7^2 < 1); only origins
  // outside that box evaluate F(0) exactly.
  // (`box_rho2` = (R + 0.7 r)^2, `box_y2` = (0.7 r)^2: wave-uniform, prepared on the host)
  bool origin_inside = false;
  if (SIDE < 0) origin_inside = (fma(Ax, Ax, Az * Az) < box_rho2) && (Ay * Ay < box_y2);
  if (!origin_inside) {
    double F0, dF0;
    torus_F<SIDE>(R, r2, Ax, Ay, Az, ux, uy, uz, F0, dF0);
    origin_inside = F0 < 0.0;
    // origin outside the body and moving away from it: both roots (if any) are behind the origin
    any = any && (origin_inside || dF0 < 0.0);
  }
  // exit root: start just outside the spheroid exit, walk left
  double t_out = ts2;
  bool pre = false;
  if (SIDE < 0) {
    // First a Newton step on the EXPANDED quartic, scaled by 1/(4R^2):  g(t) = (W/2R)^2 - rho^2,  W = |P|^2 + R^2 - r^2
    // = t^2 + 2 b t + c0,  rho^2 = al t^2 + 2 be t + ga  -- two quadratics in t, no square root: half the instructions
    // of a step on H.  Its value carries the cancellation the reference's quartic suffers from (~1e-10 mm in t), which
    // is irrelevant for a step that only has to bring the start (~1e-2 mm off) into the range where ONE step on H
    // finishes.  Near the root g = H * [(rho+R)^2 + y^2 - r^2] / 4R^2 with the second factor positive and nearly
    // constant, so the step is Newton's on H to first order; it is taken only if it moves left and stays right of the
    // closest approach, and if the iteration on H then fails from it, the proven start is used after all (below).
    const double al = fma(ux, ux, uz * uz), be = fma(Ax, ux, Az * uz), ga = fma(Ax, Ax, Az * Az);
    const double b2 = 2.0 * fma(Ay, uy, be), c0 = fma(Ay, Ay, ga) + dif2;
    const double s1 = ts2 + b2;
    const double w = fma(ts2, s1, c0) * i2R, wp = (s1 + ts2) * i2R;
    const double q1 = fma(al, ts2, be + be);
    const double Q = fma(q1, ts2, ga), Qp = fma(al, ts2, q1);
    const double g = fma(w, w, -Q), gp = fma(w + w, wp, -Qp);
    const double dt = g * rcp_seed(gp);
    pre = any && (dt > 0.0) && (dt <= ts2 - tm);
    t_out = pre ? ts2 - dt : ts2;
  }
  bool has_out = torus_newton<SIDE>(R, r2, Ax, Ay, Az, ux, uy, uz, +1.0, any, t_out);
  if (SIDE < 0) {
    const bool redo = pre && !has_out;
    if (ART_WAVE_ANY(redo)) {
      double t2 = ts2;
      const bool h2 = torus_newton<SIDE>(R, r2, Ax, Ay, Az, ux, uy, uz, +1.0, redo, t2);
      t_out = redo ? t2 : t_out;
      has_out = has_out || h2;
    }
  }
  // entry root only when the origin is outside the body: start at max(ts1, 0), walk right
  double t_in = (ts1 > 0.0 ? ts1 : 0.0);
  const bool has_in = torus_newton<SIDE>(R, r2, Ax, Ay, Az, ux, uy, uz, -1.0, any && has_out && !origin_inside, t_in);
  // entry root first: ClosestPoint lets the LATER of two equally close candidates win
  ta = t_in; tb = t_out;
  return (has_in ? 1 : 0) | (has_out ? 2 : 0);
}

// ---------------------------------------------------------------------------------------------------------
// Candidate bookkeeping of `_get_intersection`: KeepPositiveSolution (ModuleGeometry.py:110-120), the side-of-surface
// rule and support test of each mirror class, then _IntersectionRayMirror (ModuleMirror.py:27-38) with ClosestPoint
// (ModuleGeometry.py:138-147): one accepted candidate -> it, two -> the closer one (a strictly closer earlier
// candidate wins, otherwise the later), none or more than two -> the ray is lost.  Written with selects so that the
// roots can be folded in as they are found, without an indexed array.
struct Candidates {
  int cnt;
  double best;
};

template <int KIND>
ART_HD void consider(const ArtElementDesc& e, double Ax, double Ay, double Az, double ux, double uy, double uz,
                     double t, bool valid, Candidates& c) {
  if (!ART_WAVE_ANY(valid)) return;  // nothing to test in this wavefront (e.g. no entry root: origin inside the tube)
  const double x = fma(t, ux, Ax), y = fma(t, uy, Ay), z = fma(t, uz, Az);
  bool ok = valid && (t > 1e-12);
  if (KIND == ART_SPHERE || KIND == ART_CYLINDER) ok = ok && (z < 0.0) && include_support(e.support_kind, e.sp, x, y);  // :175, :841
  else if (KIND == ART_PARABOLA) ok = ok && include_support(e.support_kind, e.sp, x - e.centre[0], y - e.centre[1]);     // :344
  else if (KIND == ART_ELLIPSOID) ok = ok && (z < 0.0) && include_support(e.support_kind, e.sp, x - e.centre[0], y - e.centre[1]);  // :678
  else ok = ok && (z < -e.mp[0]) && include_support(e.support_kind, e.sp, x, y);                                           // :473
  const bool take = ok && (c.cnt == 0 || !(c.best < t));
  c.best = take ? t : c.best;
  c.cnt += ok ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------
// `_get_intersection` of the undeformed optic in the optic frame.  A = origin, u = unit direction.
// Returns hit and the ray parameter t (P = A + t u).
template <int KIND>
ART_HD bool intersect(const ArtElementDesc& e, double Ax, double Ay, double Az, double ux, double uy, double uz,
                      double& t_hit) {
  if (KIND == ART_PLANE || KIND == ART_MASK) {
    // ModuleMirror.py:73-82 (t > 0, no epsilon) ; ModuleMask.py:51-61 (passes where the support is NOT hit)
    const double t = -Az / uz;
    const bool inside = include_support(e.support_kind, e.sp, fma(t, ux, Ax), fma(t, uy, Ay));
    t_hit = t;
    return (t > 0.0) && (KIND == ART_PLANE ? inside : !inside);
  }
  Candidates c = {0, 0.0};
  if (KIND == ART_TORUS) {
    const double R = e.mp[0], r2 = e.ART_D_R2;
    double ta = 0.0, tb = 0.0;
    int n = torus_body_roots<-1>(R, r2, e.ART_D_RB, e.mp[2], e.mp[3], e.ART_D_BOX_RHO2, e.ART_D_BOX_Y2, e.ART_D_DIF2,
                                 e.ART_D_I2R, Ax, Ay, Az, ux, uy, uz, ta, tb);
    consider<KIND>(e, Ax, Ay, Az, ux, uy, uz, ta, (n & 1) != 0, c);
    consider<KIND>(e, Ax, Ay, Az, ux, uy, uz, tb, (n & 2) != 0, c);
    if (e.flags & ART_FLAG_D_LEMON) {  // self-intersecting torus (r > R): the quartic's second factor has real roots too
      // bounding sphere of the lemon: (rho + R)^2 + y^2 <= r^2  =>  rho^2 + y^2 <= r^2 - R^2 (its tips sit on the axis
      // at |y| = sqrt(r^2 - R^2), farther out than its equator rho = r - R)
      n = torus_body_roots<+1>(R, r2, sqrt(r2 - R * R), 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, Ax, Ay, Az, ux, uy, uz, ta, tb);
      consider<KIND>(e, Ax, Ay, Az, ux, uy, uz, ta, (n & 1) != 0, c);
      consider<KIND>(e, Ax, Ay, Az, ux, uy, uz, tb, (n & 2) != 0, c);
    }
  } else {
    double qa, qb, qc;
    if (KIND == ART_SPHERE) {  // :163-170
      qa = dot3(ux, uy, uz, ux, uy, uz);
      qb = 2.0 * dot3(ux, uy, uz, Ax, Ay, Az);
      qc = dot3(Ax, Ay, Az, Ax, Ay, Az) - e.mp[0] * e.mp[0];
    } else if (KIND == ART_PARABOLA) {  // :334-336
      const double p = e.mp[0];
      qa = fma(ux, ux, uy * uy);
      qb = 2.0 * fma(ux, Ax, uy * Ay) - 2.0 * p * uz;
      qc = fma(Ax, Ax, Ay * Ay) - 2.0 * p * Az;
    } else if (KIND == ART_ELLIPSOID) {  // :667-669
      const double ia2 = rcp_full(e.mp[0] * e.mp[0]), ib2 = rcp_full(e.mp[1] * e.mp[1]);
      qa = fma(uy, uy, uz * uz) * ib2 + ux * ux * ia2;
      qb = 2.0 * (fma(uy, Ay, uz * Az) * ib2 + ux * Ax * ia2);
      qc = fma(Ay, Ay, Az * Az) * ib2 + Ax * Ax * ia2 - 1.0;
    } else {  // cylinder :831-833
      qa = fma(uy, uy, uz * uz);
      qb = 2.0 * fma(uy, Ay, uz * Az);
      qc = fma(Ay, Ay, Az * Az) - e.mp[0] * e.mp[0];
    }
    double t1 = 0.0, t2 = 0.0;
    const int n = quadratic_roots(qa, qb, qc, t1, t2);
    consider<KIND>(e, Ax, Ay, Az, ux, uy, uz, t1, n > 0, c);
    consider<KIND>(e, Ax, Ay, Az, ux, uy, uz, t2, n > 1, c);
  }
  t_hit = c.best;
  return c.cnt == 1 || c.cnt == 2;
}

// Derived constants.  Every entry point works on its OWN copy of the caller's descriptors and runs prepare_element()
// on it first; the copy's `bwd` and `pos` (which the kernels do not need once the two offsets below exist: the way
// back is the transpose of `fwd`) and mp[2..3] of a torus then hold:
//   bwd[0..2] = centre - fwd * pos       P_optic = fwd * P_lab + bwd[0..2]
//   bwd[3..5] = pos - fwd^T * centre     P_lab   = fwd^T * P_optic + bwd[3..5]
// (formed in long double, so that they carry half an ulp of their own and nothing of the products' cancellation) and,
// for a torus, the wave-uniform constants its intersection and normal would otherwise recompute in every lane:
//   mp[2] = 1/(R+r)^2, mp[3] = 1/(r(R+r))  (spheroid of torus_body_roots)   bwd[6] = r^2
//   bwd[7] = (R + 0.7 r)^2, bwd[8] = (0.7 r)^2  (origin-inside box)         pos[0] = R + r, pos[1] = R^2 + r^2,
//   pos[2] = R^2 - r^2, mp[1] = 1/(2R) (r itself is not needed any more), flags bit 31 = (r > R).
inline void prepare_element(ArtElementDesc& e) {
  long double in[3], out[3];
  for (int i = 0; i < 3; ++i) {
    in[i] = (long double)e.centre[i];
    out[i] = (long double)e.pos[i];
    for (int j = 0; j < 3; ++j) {
      in[i] -= (long double)e.fwd[3 * i + j] * (long double)e.pos[j];
      out[i] -= (long double)e.fwd[3 * j + i] * (long double)e.centre[j];
    }
  }
  for (int i = 0; i < 3; ++i) { e.bwd[i] = (double)in[i]; e.bwd[3 + i] = (double)out[i]; }
  e.bwd[6] = e.bwd[7] = e.bwd[8] = 0.0;
  e.pos[0] = e.pos[1] = e.pos[2] = 0.0;
  e.flags &= ~ART_FLAG_D_LEMON;
  if (e.kind == ART_TORUS) {
    const double R = e.mp[0], r = e.mp[1];
    e.mp[2] = 1.0 / ((R + r) * (R + r));
    e.mp[3] = 1.0 / (r * (R + r));
    e.ART_D_R2 = r * r;
    e.ART_D_BOX_RHO2 = (R + 0.7 * r) * (R + 0.7 * r);
    e.ART_D_BOX_Y2 = (0.7 * r) * (0.7 * r);
    e.ART_D_RB = R + r;
    e.ART_D_SUM2 = R * R + r * r;
    e.ART_D_DIF2 = R * R - r * r;
    e.ART_D_I2R = 0.5 / R;
    if (r > R) e.flags |= ART_FLAG_D_LEMON;
  }
}

// ---------------------------------------------------------------------------------------------------------
// KIND >= 0: compile-time optic kind; KIND = ART_KIND_DYN: wave-uniform run-time switch on e.kind.  The run-time form
// lets the fused chain kernel WITH defects share one copy of everything that does not depend on the kind (frame
// changes, the unrolled Zernike evaluators, reflection, incidence angle) instead of one copy per kind.
#define ART_KIND_DYN (-1)

template <int KIND>
ART_HD bool intersect_k(const ArtElementDesc& e, double Ax, double Ay, double Az, double ux, double uy, double uz,
                        double& t) {
  if (KIND != ART_KIND_DYN) return intersect<(KIND < 0 ? 0 : KIND)>(e, Ax, Ay, Az, ux, uy, uz, t);
  switch (e.kind) {
    case ART_PLANE: return intersect<ART_PLANE>(e, Ax, Ay, Az, ux, uy, uz, t);
    case ART_SPHERE: return intersect<ART_SPHERE>(e, Ax, Ay, Az, ux, uy, uz, t);
    case ART_PARABOLA: return intersect<ART_PARABOLA>(e, Ax, Ay, Az, ux, uy, uz, t);
    case ART_TORUS: return intersect<ART_TORUS>(e, Ax, Ay, Az, ux, uy, uz, t);
    case ART_ELLIPSOID: return intersect<ART_ELLIPSOID>(e, Ax, Ay, Az, ux, uy, uz, t);
    case ART_CYLINDER: return intersect<ART_CYLINDER>(e, Ax, Ay, Az, ux, uy, uz, t);
    default: return intersect<ART_MASK>(e, Ax, Ay, Az, ux, uy, uz, t);
  }
}

template <int KIND>
ART_HD void base_normal_k(const ArtElementDesc& e, double x, double y, double z, double& nx, double& ny, double& nz) {
  if (KIND != ART_KIND_DYN) return base_normal<(KIND < 0 ? 0 : KIND)>(e, x, y, z, nx, ny, nz);
  switch (e.kind) {
    case ART_SPHERE: return base_normal<ART_SPHERE>(e, x, y, z, nx, ny, nz);
    case ART_PARABOLA: return base_normal<ART_PARABOLA>(e, x, y, z, nx, ny, nz);
    case ART_TORUS: return base_normal<ART_TORUS>(e, x, y, z, nx, ny, nz);
    case ART_ELLIPSOID: return base_normal<ART_ELLIPSOID>(e, x, y, z, nx, ny, nz);
    case ART_CYLINDER: return base_normal<ART_CYLINDER>(e, x, y, z, nx, ny, nz);
    default: return base_normal<ART_PLANE>(e, x, y, z, nx, ny, nz);
  }
}

// ---------------------------------------------------------------------------------------------------------
// One element acting on one ray (ART/ModuleProcessing.py:284-311 for a single ray).
// Returns false when the ray is lost (missed the optic / blocked by the mask).
// `zern`: the element's dense Zernike tables (n_defects x ART_ZERN_STRIDE doubles) -- e.zern itself (device memory,
// read through scalar loads) or, in the -DART_ZERN_LDS build, their copy in LDS.
template <int KIND, bool DEFECT, bool ZGEN = false>
ART_HD bool trace_ray(const ArtElementDesc& e, const double* zern, Ray& r) {
  // lab -> optic frame (:289-295)
  double Ax, Ay, Az, ux, uy, uz;
  mat3_apply_off(e.fwd, e.ART_D_IN_OFF, r.ox, r.oy, r.oz, Ax, Ay, Az);
  mat3_apply(e.fwd, r.dx, r.dy, r.dz, ux, uy, uz);

  double t;
  if (!intersect_k<KIND>(e, Ax, Ay, Az, ux, uy, uz, t)) return false;
  double Px = fma(t, ux, Ax), Py = fma(t, uy, Ay), Pz = fma(t, uz, Az);

  double vx, vy, vz, inc;
  if (KIND == ART_MASK || (KIND == ART_KIND_DYN && e.kind == ART_MASK)) {
    // _TransmitMaskRay, ModuleMask.py:93-108: direction unchanged, incidence = angle(v, ez)
    vx = ux; vy = uy; vz = uz;
    inc = kahan_angle_unit(ux, uy, uz, 0.0, 0.0, 1.0, uz);
  } else {
    double nx, ny, nz;
#ifdef ART_DIAG_NONORMAL
    nx = 0.0; ny = 0.0; nz = 1.0 + 1e-30 * Px;
#else
    base_normal_k<KIND>(e, Px, Py, Pz, nx, ny, nz);
#endif
    if (DEFECT && (e.n_defects > 0 || e.n_grid > 0)) {
      // DeformedMirror._get_intersection, ModuleMirror.py:969-980: slide the hit point along the ray by
      // h / cos(alpha), h = summed defect offsets at (P - centre), alpha = angle(-u, base normal)
      double h = 0.0;
      if (ZGEN) {       // tables in the recurrence layout, all of one order (padded by the host)
        const int zs = zgen_stride((int)zern[1]);
        for (int d = 0; d < e.n_defects; ++d) {
          double hd, g1, g2;
          zernike_recurrence(zern + d * zs, Px - e.centre[0], Py - e.centre[1], false, hd, g1, g2);
          h += hd;
        }
      } else {
        for (int d = 0; d < e.n_defects; ++d)
          h += zernike_offset(ART_ZUNI(zern + d * ART_ZERN_STRIDE), Px - e.centre[0], Py - e.centre[1]);
      }
      for (int d = 0; d < e.n_grid; ++d) h += grid_offset(e.grid[d], Px - e.centre[0], Py - e.centre[1]);
      const double cosa = -dot3(ux, uy, uz, nx, ny, nz);
      const double s = div_full(h, cosa);
      t -= s;
      Px = fma(-s, ux, Px); Py = fma(-s, uy, Py); Pz = fma(-s, uz, Pz);
      base_normal_k<KIND>(e, Px, Py, Pz, nx, ny, nz);
      if (e.flags & ART_FLAG_PERTURBED_NORMAL) {
        // DeformedMirror.get_normal, ModuleMirror.py:952-961 with normal_add (ModuleGeometry.py:394-407):
        // surface slopes add up
        const double inz = rcp_full(nz);
        double gXs = -nx * inz, gYs = -ny * inz;
        for (int d = 0; d < e.n_defects; ++d) {
          double gX, gY;
          if (ZGEN) {
            double hd;
            zernike_recurrence(zern + d * zgen_stride((int)zern[1]), Px - e.centre[0], Py - e.centre[1], true, hd, gX, gY);
          } else {
            zernike_slopes(ART_ZUNI(zern + d * ART_ZERN_STRIDE), Px - e.centre[0], Py - e.centre[1], gX, gY);
          }
          gXs += gX; gYs += gY;
        }
        const double inv = rsqrt_full(fma(gXs, gXs, fma(gYs, gYs, 1.0)));
        nx = -gXs * inv; ny = -gYs * inv; nz = inv;
      }
    }
    // _ReflectionMirrorRay, ModuleMirror.py:878-906: v' = rot(n, pi) applied to -v  ==  v - 2 (n.v) n
    const double dn = dot3(ux, uy, uz, nx, ny, nz);
    vx = fma(-2.0 * dn, nx, ux); vy = fma(-2.0 * dn, ny, uy); vz = fma(-2.0 * dn, nz, uz);
#ifdef ART_DIAG_NOINC
    inc = dn;
#else
    inc = kahan_angle_unit(-ux, -uy, -uz, nx, ny, nz, -dn);
#endif
  }
  // optic -> lab frame (:306-309)
  double dx, dy, dz;
  // (a caller's bwd holds the same map built the reference's way; the kernels use the transpose of fwd, which is
  // equal to it to rounding, to halve the constants they keep in scalar registers)
  mat3t_apply_off(e.fwd, e.ART_D_OUT_OFF, Px, Py, Pz, r.ox, r.oy, r.oz);
  mat3t_apply(e.fwd, vx, vy, vz, dx, dy, dz);
  // Ray.vector setter renormalises (ModuleOpticalRay.py:85-90): one Newton step of 1/sqrt on |d|^2 ~ 1
  const double s2 = dot3(dx, dy, dz, dx, dy, dz);
  const double k = fma(-0.5, s2, 1.5);
  r.dx = dx * k; r.dy = dy * k; r.dz = dz * k;
  r.path += t;  // |P - A| with |u| = 1 (ModuleMirror.py:904, ModuleMask.py:100)
  r.inc = inc;
  return true;
}

// runtime dispatch on the optic kind (wave-uniform).  Without defects: one fully specialised body per kind (the
// headline path; registers and schedule as tuned in round 1).  With defects: ONE body whose kind-dependent parts
// (intersection, undeformed normal) switch at run time -- a sixth of the code, and the register allocator no longer
// has to satisfy six inlined copies of the unrolled Zernike evaluators at once.
template <bool DEFECT>
ART_HD bool trace_ray_dyn(const ArtElementDesc& e, const double* zern, Ray& r) {
#if !defined(__HIP_DEVICE_COMPILE__)
  // (host twin: one entry point for everything; on the device the recurrence path has its own kernel)
  if (DEFECT && (e.flags & ART_FLAG_ZERN_RECURRENCE)) return trace_ray<ART_KIND_DYN, true, true>(e, zern, r);
#endif
  if (DEFECT) return trace_ray<ART_KIND_DYN, true>(e, zern, r);
  switch (e.kind) {
    case ART_PLANE: return trace_ray<ART_PLANE, false>(e, zern, r);
    case ART_SPHERE: return trace_ray<ART_SPHERE, false>(e, zern, r);
    case ART_PARABOLA: return trace_ray<ART_PARABOLA, false>(e, zern, r);
    case ART_TORUS: return trace_ray<ART_TORUS, false>(e, zern, r);
    case ART_ELLIPSOID: return trace_ray<ART_ELLIPSOID, false>(e, zern, r);
    case ART_CYLINDER: return trace_ray<ART_CYLINDER, false>(e, zern, r);
    default: return trace_ray<ART_MASK, false>(e, zern, r);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Detector read-out for one ray (ART/ModuleDetector.py:191-234, :272-275; ModuleGeometry.py:48-57)
ART_HD void detector_ray(const ArtDetectorDesc& d, const Ray& r, double& Ix, double& Iy, double& Iz, double& X,
                         double& Y, double& opl) {
  const double num = dot3(d.normal[0], d.normal[1], d.normal[2], d.centre[0] - r.ox, d.centre[1] - r.oy,
                          d.centre[2] - r.oz);
  const double den = dot3(r.dx, r.dy, r.dz, d.normal[0], d.normal[1], d.normal[2]);
  // ordinary magnitudes (mm-scale lengths over a direction cosine): the refined-seed division and square root of
  // the tracing path, a third of the IEEE expansions.  A ray parallel to the detector (den = 0) reads NaN.
  const double t = div_full(num, den);
  Ix = fma(t, r.dx, r.ox); Iy = fma(t, r.dy, r.oy); Iz = fma(t, r.dz, r.oz);
  double rx, ry, rz;
  mat3_apply(d.rot, Ix - d.centre[0], Iy - d.centre[1], Iz - d.centre[2], rx, ry, rz);
  X = rx; Y = ry;
  double un, iun;
  sqrt_rsqrt_coarse(dot3(r.dx, r.dy, r.dz, r.dx, r.dy, r.dz), un, iun);
  opl = fma(fabs(t), un, r.path);
}

// Read-out at the current detector position and its derivative with respect to a shift s along -normal
// (Detector.shiftByDistance(s), ART/ModuleDetector.py:163-177): everything is linear in s.
// `crosses`: does the ray's hit change side of its origin (t changes sign) for some shift in [0, span]?  There
// opl = |t| |u| + path has a kink and the linear model only describes shift 0.
ART_HD void detector_ray_scan(const ArtDetectorDesc& d, const Ray& r, double span, double& X, double& Y, double& opl,
                              double& sx, double& sy, double& so, bool& crosses) {
  double Ix, Iy, Iz;
  detector_ray(d, r, Ix, Iy, Iz, X, Y, opl);
  const double den = dot3(r.dx, r.dy, r.dz, d.normal[0], d.normal[1], d.normal[2]);
  const double nn = dot3(d.normal[0], d.normal[1], d.normal[2], d.normal[0], d.normal[1], d.normal[2]);
  const double dt = -nn / den;  // d t / d s: the plane recedes by s*normal, t = n.(C - s n - A)/(u.n)
  // d(I - C_s)/ds = dt*u + normal
  double rx, ry, rz;
  mat3_apply(d.rot, fma(dt, r.dx, d.normal[0]), fma(dt, r.dy, d.normal[1]), fma(dt, r.dz, d.normal[2]), rx, ry, rz);
  sx = rx; sy = ry;
  const double num = dot3(d.normal[0], d.normal[1], d.normal[2], d.centre[0] - r.ox, d.centre[1] - r.oy,
                          d.centre[2] - r.oz);
  const double t0 = num / den;
  const double sgn = (t0 >= 0.0) ? 1.0 : -1.0;  // opl = |t| |u| + path
  so = sgn * dt * sqrt(dot3(r.dx, r.dy, r.dz, r.dx, r.dy, r.dz));
  crosses = t0 * fma(span, dt, t0) < 0.0;
}

// The same for art_analyse_bundles, with the shift at which the hit point passes through the ray's origin instead of a
// yes/no for a given span (t(s) = t0 + s dt vanishes at s_kink = -t0 / dt = (n.(C - A)) / (n.n): no division by u.n), and
// |u|, which the caller needs again.  ONE reciprocal of u.n serves t0, dt and the slopes; (X, Y, opl) at shift 0 are
// detector_ray's to rounding (t = num * (1 / den) with one residual correction, as div_full does).
ART_HD void detector_ray_scan_kink(const ArtDetectorDesc& d, const Ray& r, double inv_nn, double& X, double& Y, double& opl,
                                   double& sx, double& sy, double& so, double& s_kink, double& un) {
  const double num = dot3(d.normal[0], d.normal[1], d.normal[2], d.centre[0] - r.ox, d.centre[1] - r.oy,
                          d.centre[2] - r.oz);
  const double den = dot3(r.dx, r.dy, r.dz, d.normal[0], d.normal[1], d.normal[2]);
  const double iden = rcp_full(den);
  const double q = num * iden;
  const double t = fma(fma(-den, q, num), iden, q);          // = div_full(num, den)
  const double Ix = fma(t, r.dx, r.ox), Iy = fma(t, r.dy, r.oy), Iz = fma(t, r.dz, r.oz);
  double rx, ry, rz;
  mat3_apply(d.rot, Ix - d.centre[0], Iy - d.centre[1], Iz - d.centre[2], rx, ry, rz);
  X = rx; Y = ry;
  double iun;
  sqrt_rsqrt_coarse(dot3(r.dx, r.dy, r.dz, r.dx, r.dy, r.dz), un, iun);
  opl = fma(fabs(t), un, r.path);
  const double nn = dot3(d.normal[0], d.normal[1], d.normal[2], d.normal[0], d.normal[1], d.normal[2]);
  const double dt = -nn * iden;                               // d t / d s: the plane recedes by s * normal
  mat3_apply(d.rot, fma(dt, r.dx, d.normal[0]), fma(dt, r.dy, d.normal[1]), fma(dt, r.dz, d.normal[2]), rx, ry, rz);
  sx = rx; sy = ry;
  so = ((t >= 0.0) ? dt : -dt) * un;                          // opl = |t| |u| + path
  s_kink = (den != 0.0) ? num * inv_nn : INFINITY;
}

// ---------------------------------------------------------------------------------------------------------
// Detector placement on the device (art_analyse_bundles; ART/ModuleDetector.py:109-137, ART/ModuleGeometry.py:321-343).
// 3x3 map of RotationPoint(., a, ez) for a UNIT vector a, with the reference's special cases: identity for an angle
// below 1e-10, the point inversion -I within 1e-10 of pi; otherwise the rotation by the Kahan angle about a x ez, written
// as the Rodrigues sum with half-angle terms that the host shell uses (ModuleGeometry.RotationAroundAxis).
ART_HD void rotation_to_ez(const double* a, double* M) {
  const double ang = kahan_angle_unit(a[0], a[1], a[2], 0.0, 0.0, 1.0, a[2]);
  const bool same = fabs(ang) < 1e-10, opposite = fabs(ang - 3.14159265358979323846) < 1e-10;
  if (same || opposite) {
    const double d = same ? 1.0 : -1.0;
    M[0] = d; M[1] = 0.0; M[2] = 0.0; M[3] = 0.0; M[4] = d; M[5] = 0.0; M[6] = 0.0; M[7] = 0.0; M[8] = d;
    return;
  }
  // axis = a x ez = (a1, -a0, 0), normalised; q = sin(ang/2) axis, w = cos(ang/2)
  const double nx = a[1], ny = -a[0];
  const double nrm = sqrt(fma(nx, nx, ny * ny));
  const double sh = sin(0.5 * ang), w = cos(0.5 * ang);
  const double qx = sh * (nx / nrm), qy = sh * (ny / nrm), qz = 0.0;
  for (int j = 0; j < 3; ++j) {      // column j = image of the basis vector e_j: v + 2 (w (q x v) + q x (q x v))
    const double vx = (j == 0) ? 1.0 : 0.0, vy = (j == 1) ? 1.0 : 0.0, vz = (j == 2) ? 1.0 : 0.0;
    const double cx = qy * vz - qz * vy, cy = qz * vx - qx * vz, cz = qx * vy - qy * vx;
    const double ex = qy * cz - qz * cy, ey = qz * cx - qx * cz, ez = qx * cy - qy * cx;
    M[0 + j] = vx + 2.0 * (w * cx + ex);
    M[3 + j] = vy + 2.0 * (w * cy + ey);
    M[6 + j] = vz + 2.0 * (w * cz + ez);
  }
}

// sums9 = count, sum point (3), sum vector (3), sum w, sum path.  Fills the detector (centre, normal, rot), its
// reference point, the mean direction (unit) and the provisional path centre co.  mode: ArtJobMode (0 autoplace, 1 manual).
// The sequence of normalisations is the reference's: FindCentralRay's mean vector goes through the Ray.vector setter
// (ART/ModuleOpticalRay.py:85-90), its negative through the Detector.normal setter (ART/ModuleDetector.py:57-66).
ART_HD void analysis_place(const double* sums9, int mode, double distance, const double* centre_in,
                           const double* normal_in, const double* refpoint_in, ArtDetectorDesc& d, double* refpoint,
                           double* axis, double& co) {
  const double cnt = sums9[0];
  const double px = sums9[1] / cnt, py = sums9[2] / cnt, pz = sums9[3] / cnt;
  double vx = sums9[4] / cnt, vy = sums9[5] / cnt, vz = sums9[6] / cnt;
  const double vl = sqrt(dot3(vx, vy, vz, vx, vy, vz));
  vx /= vl; vy /= vl; vz /= vl;
  axis[0] = vx; axis[1] = vy; axis[2] = vz;
  if (mode == 1) {
    for (int k = 0; k < 3; ++k) {
      d.normal[k] = normal_in[k];       // unit by contract (ArtDetectorDesc everywhere): used bit for bit
      d.centre[k] = centre_in[k];
      refpoint[k] = refpoint_in[k];
    }
  } else {
    const double nx = -vx, ny = -vy, nz = -vz;
    const double nl = sqrt(dot3(nx, ny, nz, nx, ny, nz));
    d.normal[0] = nx / nl; d.normal[1] = ny / nl; d.normal[2] = nz / nl;
    d.centre[0] = px - d.normal[0] * distance;
    d.centre[1] = py - d.normal[1] * distance;
    d.centre[2] = pz - d.normal[2] * distance;
    refpoint[0] = px; refpoint[1] = py; refpoint[2] = pz;
  }
  rotation_to_ez(d.normal, d.rot);
  // the mean ray's distance to the detector plane: within a fraction of a millimetre of the mean of |I - A| over the rays
  const double num = dot3(d.normal[0], d.normal[1], d.normal[2], d.centre[0] - px, d.centre[1] - py, d.centre[2] - pz);
  const double den = dot3(vx, vy, vz, d.normal[0], d.normal[1], d.normal[2]);
  co = sums9[8] / cnt + fabs(num / den);
}

// ---------------------------------------------------------------------------------------------------------
// Sources (ART/ModuleSource.py:23-81, :135-169; SpiralVogel ART/ModuleGeometry.py:61-76) for ray index k
ART_HD void vogel_point(int64_t k, int64_t n_total, double radius, double& x, double& y) {
  const double golden = 3.14159265358979323846 * (3.0 - sqrt(5.0));
  const double rr = sqrt((double)k / (double)n_total) * radius;
  const double theta = golden * (double)k;
  x = cos(theta) * rr;
  y = sin(theta) * rr;
}

// (px,py,0) + direction (vx,vy,vz) in the source frame -> lab frame
ART_HD void source_place(const double* rot, const double* S, double px, double py, double vx, double vy, double vz,
                         Ray& r) {
  double ox, oy, oz, dx, dy, dz;
  mat3_apply(rot, px, py, 0.0, ox, oy, oz);
  mat3_apply(rot, vx, vy, vz, dx, dy, dz);
  r.ox = ox + S[0]; r.oy = oy + S[1]; r.oz = oz + S[2];
  r.dx = dx; r.dy = dy; r.dz = dz;
  r.path = 0.0;
  r.inc = NAN;
}

ART_HD void source_ray(int kind, double size, const double* rot, const double* S, int64_t k, int64_t n_total,
                       Ray& r) {
  double x, y;
  vogel_point(k, n_total, (kind == 0) ? tan(size) : size, x, y);  // _Cone: Height = 1, Radius = tan(Angle)
  if (kind == 0) {
    const double inv = 1.0 / sqrt(fma(x, x, fma(y, y, 1.0)));
    source_place(rot, S, 0.0, 0.0, x * inv, y * inv, inv, r);
  } else {
    source_place(rot, S, x, y, 0.0, 0.0, 1.0, r);
  }
}

// ExtendedSource (ART/ModuleSource.py:85-131): `n_points` point sources on a Vogel disk of radius `radius`, each
// emitting the same cone of `per` rays; ray number = point * per + ray-in-cone (:124-127).
ART_HD void source_ray_extended(double radius, double divergence, int64_t n_points, int64_t per, const double* rot,
                                const double* S, int64_t k, Ray& r) {
  const int64_t point = k / per, l = k - point * per;
  double px, py, x, y;
  vogel_point(point, n_points, radius, px, py);
  vogel_point(l, per, tan(divergence), x, y);
  const double inv = 1.0 / sqrt(fma(x, x, fma(y, y, 1.0)));
  source_place(rot, S, px, py, x * inv, y * inv, inv, r);
}

}  // namespace art
