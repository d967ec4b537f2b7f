// ba_math.h — per-observation maths of the bundle-adjustment hot path.
//
// These inline functions are the bodies of the HIP kernels in kernels.hip. They
// are marked __host__ __device__ so that (a) the host LM driver can evaluate the
// handful of residual blocks that never reach the device (all-constant "fixed
// cost" blocks) with the same arithmetic and (b) tests/ can compile this header
// with g++ and compare it with the oracle without a GPU. There is no CPU solver
// path: nothing here iterates over a problem.
//
// What is evaluated (reference file:line, /root/reference):
//   residual  r = world2image(R(rvec) X + t; intrinsics) - uv
//             src/base3d/bundle_adjustment.h:131-159 (BACostFunction<Model>::operator())
//   models    src/base3d/camera_models.h:111-130 (PINHOLE), :170-193 + :225-242 (OPENCV),
//             :277-302 + :340-357 (CATA)
//   Jacobian  the reference gets d r / d(rvec,tx,ty,tz,X,intrinsics) from ceres Jets
//             (bundle_adjustment.h:124-129); here it is the closed form:
//               d r/d t = A,  d r/d X = A R,  d r/d rvec = -A [R X]x Jl(rvec),  A = dC/dXc
//   loss      ceres::CauchyLoss(a) + Corrector with rho'' <= 0 (bundle_adjustment.cc:477-478)
//   prior     BARotationConstraintCostFunction, bundle_adjustment.cc:72-111
#ifndef MAVBA_BA_MATH_H_
#define MAVBA_BA_MATH_H_

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MAVBA_HD __host__ __device__ __forceinline__
#else
#define MAVBA_HD inline
#endif

#define MAVBA_M_PINHOLE 1
#define MAVBA_M_OPENCV 2
#define MAVBA_M_CATA 3

namespace mavba {

// ---------------------------------------------------------------------------
// Camera record: what a pose costs per observation once the transcendental
// part is hoisted out. rec = { w[3], t[3], a, b, c } with
//   R(w)  = I + a [w]x + b [w]x^2          a = sin(th)/th, b = (1-cos th)/th^2
//   Jl(w) = I + b [w]x + c [w]x^2          c = (th - sin th)/th^3
// 9 doubles = 72 B per image, so 2000 images fit the 160 KB LDS of one CU.
// ---------------------------------------------------------------------------
MAVBA_HD void rot_coeffs(double th2, double& a, double& b, double& c) {
  if (th2 > 1e-6) {
    const double th = sqrt(th2);
    const double s = sin(th), co = cos(th);
    a = s / th;
    b = (1.0 - co) / th2;
    c = (th - s) / (th2 * th);
  } else {  // series; truncation error < th^6/5040 < 2e-22
    a = 1.0 - th2 * (1.0 / 6.0 - th2 / 120.0);
    b = 0.5 - th2 * (1.0 / 24.0 - th2 / 720.0);
    c = 1.0 / 6.0 - th2 * (1.0 / 120.0 - th2 / 5040.0);
  }
}

MAVBA_HD void cam_prepare(const double* pose, double* rec) {
  const double wx = pose[0], wy = pose[1], wz = pose[2];
  double a, b, c;
  rot_coeffs(wx * wx + wy * wy + wz * wz, a, b, c);
  rec[0] = wx; rec[1] = wy; rec[2] = wz;
  rec[3] = pose[3]; rec[4] = pose[4]; rec[5] = pose[5];
  rec[6] = a; rec[7] = b; rec[8] = c;
}

// 1 / x and 1 / sqrt(x) per observation (round 5). The IEEE division and sqrt + division of the device build are 20-35
// FP64 instructions each on a pipe that every per-observation kernel is bound by; the hardware seeds (v_rcp_f64 / v_rsq_f64,
// ~2^-24 relative) with two Newton steps / one third-order step are 5-6 and end within 1-2 ulp (measured for the rsqrt:
// 1.4e-16 against 1.1e-16 correctly rounded, scripts/_dbg/tile_bench.hip). Host builds (tests, the fixed-cost blocks) keep
// the IEEE forms: the two agree to ~2e-16, far inside every tolerance of the parity tests.
MAVBA_HD double rcp_obs(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, e, y);
#else
  return 1.0 / x;
#endif
}
MAVBA_HD double rsqrt_obs(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(x);
  const double h = __builtin_fma(-(x * y), y, 1.0);
  return __builtin_fma(y * h, __builtin_fma(0.375, h, 0.5), y);
#else
  return 1.0 / sqrt(x);
#endif
}

// log(x) per observation, x >= 1 and finite (the Cauchy loss: x = 1 + s / b), round 6. The library's log is ~90 vector
// instructions per observation in every kernel that needs the cost (it carries a double-double tail for < 1 ulp); this one
// is 35: x = 2^e m with m in [sqrt(1/2), sqrt(2)), t = (m - 1) / (m + 1) <= 0.1716, log m = 2 t (1 + t^2/3 + ... + t^20/21)
// (the tail is below 7e-19 of the leading term), e ln 2 in two parts. Error <= 2 ulp (tests/test_host_math.py compares
// the same arithmetic on the host with log over 1 <= x < 1e12); host builds keep the library's.
// (the series, given x = 2^e m with m in [0.5, 1): host-callable for the test)
MAVBA_HD double log_from_parts(double m, int e) {
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;
  e = low ? e - 1 : e;
  const double f = m - 1.0, d = m + 1.0;
  const double r = rcp_obs(d);
  double t = f * r;
  t = __builtin_fma(r, __builtin_fma(-t, d, f), t);
  const double t2 = t * t;
  double p = 1.0 / 21.0;
  p = __builtin_fma(p, t2, 1.0 / 19.0); p = __builtin_fma(p, t2, 1.0 / 17.0); p = __builtin_fma(p, t2, 1.0 / 15.0);
  p = __builtin_fma(p, t2, 1.0 / 13.0); p = __builtin_fma(p, t2, 1.0 / 11.0); p = __builtin_fma(p, t2, 1.0 / 9.0);
  p = __builtin_fma(p, t2, 1.0 / 7.0); p = __builtin_fma(p, t2, 1.0 / 5.0); p = __builtin_fma(p, t2, 1.0 / 3.0);
  const double t3 = t * t2;
  const double ed = (double)e;
  // log m = 2 t + 2 t^3 p; e ln 2 = e hi + e lo with hi = the leading 32 bits of ln 2 (e hi is exact)
  const double lm = __builtin_fma(t3 + t3, p, t + t);
  return __builtin_fma(ed, 6.93147180369123816490e-01, __builtin_fma(ed, 1.90821492927058770002e-10, lm));
}
MAVBA_HD double log_obs(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return log_from_parts(__builtin_amdgcn_frexp_mant(x), __builtin_amdgcn_frexp_exp(x));
#else
  return log(x);
#endif
}

// out = m x w
MAVBA_HD void cross3(const double* m, const double* w, double* out) {
  out[0] = m[1] * w[2] - m[2] * w[1];
  out[1] = m[2] * w[0] - m[0] * w[2];
  out[2] = m[0] * w[1] - m[1] * w[0];
}

// Camera-frame point Xc = R X + t and the rotated part Xr = R X.
MAVBA_HD void transform_point(const double* rec, const double* X, double* Xr, double* Xc) {
  double wX[3], wwX[3];
  cross3(rec, X, wX);     // w x X
  cross3(rec, wX, wwX);   // w x (w x X)
  const double a = rec[6], b = rec[7];
  Xr[0] = X[0] + a * wX[0] + b * wwX[0];
  Xr[1] = X[1] + a * wX[1] + b * wwX[1];
  Xr[2] = X[2] + a * wX[2] + b * wwX[2];
  Xc[0] = Xr[0] + rec[3]; Xc[1] = Xr[1] + rec[4]; Xc[2] = Xr[2] + rec[5];
}

// Projection of a camera-frame point. Returns (u, v); if WANT_J also
//   A[6]   = d(u,v)/dXc (row-major 2x3)
//   Jk[18] = d(u,v)/d intrinsics, row-major 2x9 (columns >= K are zero)
template <bool WANT_J>
MAVBA_HD void project(int model, const double* cam, const double* Xc, double& u, double& v,
                      double* A, double* Jk) {
  const double fx = cam[0], fy = cam[1];
  double zz = Xc[2], nrm = 0.0;
  if (model == MAVBA_M_CATA) {
    nrm = sqrt(Xc[0] * Xc[0] + Xc[1] * Xc[1] + Xc[2] * Xc[2]);
    zz = Xc[2] + cam[8] * nrm;
  }
  const double iz = rcp_obs(zz);
  const double un = Xc[0] * iz, vn = Xc[1] * iz;
  double ud = un, vd = vn;
  double D0 = 1.0, D1 = 0.0, D2 = 0.0, D3 = 1.0;  // d(ud,vd)/d(un,vn)
  double r2 = 0.0, u2 = 0.0, v2 = 0.0, uvn = 0.0;
  if (model != MAVBA_M_PINHOLE) {
    const double k1 = cam[4], k2 = cam[5], p1 = cam[6], p2 = cam[7];
    u2 = un * un; v2 = vn * vn; uvn = un * vn; r2 = u2 + v2;
    const double radial = k1 * r2 + k2 * r2 * r2;
    ud = un + (un * radial + 2.0 * p1 * uvn + p2 * (r2 + 2.0 * u2));
    vd = vn + (vn * radial + 2.0 * p2 * uvn + p1 * (r2 + 2.0 * v2));
    if (WANT_J) {
      const double drad2 = 2.0 * (k1 + 2.0 * k2 * r2);  // d radial/d un = drad2*un, /d vn = drad2*vn
      D0 = 1.0 + radial + u2 * drad2 + 2.0 * p1 * vn + 6.0 * p2 * un;
      D1 = uvn * drad2 + 2.0 * p1 * un + 2.0 * p2 * vn;
      D2 = D1;
      D3 = 1.0 + radial + v2 * drad2 + 2.0 * p2 * un + 6.0 * p1 * vn;
    }
  }
  u = fx * ud + cam[2];
  v = fy * vd + cam[3];
  if (WANT_J) {
    // d(un,vn)/dXc
    double dz0 = 0.0, dz1 = 0.0, dz2 = 1.0;
    if (model == MAVBA_M_CATA && nrm > 0.0) {
      const double s = cam[8] / nrm;
      dz0 = s * Xc[0]; dz1 = s * Xc[1]; dz2 = 1.0 + s * Xc[2];
    }
    const double n0 = iz - un * iz * dz0, n1 = -un * iz * dz1, n2 = -un * iz * dz2;
    const double n3 = -vn * iz * dz0, n4 = iz - vn * iz * dz1, n5 = -vn * iz * dz2;
    A[0] = fx * (D0 * n0 + D1 * n3); A[1] = fx * (D0 * n1 + D1 * n4); A[2] = fx * (D0 * n2 + D1 * n5);
    A[3] = fy * (D2 * n0 + D3 * n3); A[4] = fy * (D2 * n1 + D3 * n4); A[5] = fy * (D2 * n2 + D3 * n5);
#pragma unroll
    for (int i = 0; i < 18; ++i) Jk[i] = 0.0;
    Jk[0] = ud; Jk[2] = 1.0; Jk[9 + 1] = vd; Jk[9 + 3] = 1.0;
    if (model != MAVBA_M_PINHOLE) {
      Jk[4] = fx * un * r2;           Jk[9 + 4] = fy * vn * r2;
      Jk[5] = fx * un * r2 * r2;      Jk[9 + 5] = fy * vn * r2 * r2;
      Jk[6] = fx * 2.0 * uvn;         Jk[9 + 6] = fy * (r2 + 2.0 * v2);
      Jk[7] = fx * (r2 + 2.0 * u2);   Jk[9 + 7] = fy * 2.0 * uvn;
      if (model == MAVBA_M_CATA) {
        const double dun = -un * nrm * iz, dvn = -vn * nrm * iz;  // d(un,vn)/d xi
        Jk[8] = fx * (D0 * dun + D1 * dvn);
        Jk[9 + 8] = fy * (D2 * dun + D3 * dvn);
      }
    }
  }
}

// project<true> with the intrinsics Jacobian already contracted with a step dk: tk = Jk dk (2-vector) instead of Jk[18]
// (the same expressions as above, the zero entries never formed).
MAVBA_HD void project_dk(int model, const double* cam, const double* Xc, const double* dk, double& u, double& v, double* A,
                         double& tk0, double& tk1) {
  const double fx = cam[0], fy = cam[1];
  double zz = Xc[2], nrm = 0.0;
  if (model == MAVBA_M_CATA) {
    nrm = sqrt(Xc[0] * Xc[0] + Xc[1] * Xc[1] + Xc[2] * Xc[2]);
    zz = Xc[2] + cam[8] * nrm;
  }
  const double iz = rcp_obs(zz);
  const double un = Xc[0] * iz, vn = Xc[1] * iz;
  double ud = un, vd = vn;
  double D0 = 1.0, D1 = 0.0, D2 = 0.0, D3 = 1.0;
  double su = 0.0, sv = 0.0;  // the distortion parameters' part of d(ud, vd)
  if (model != MAVBA_M_PINHOLE) {
    const double k1 = cam[4], k2 = cam[5], p1 = cam[6], p2 = cam[7];
    const double u2 = un * un, v2 = vn * vn, uvn = un * vn, r2 = u2 + v2;
    const double radial = k1 * r2 + k2 * r2 * r2;
    ud = un + (un * radial + 2.0 * p1 * uvn + p2 * (r2 + 2.0 * u2));
    vd = vn + (vn * radial + 2.0 * p2 * uvn + p1 * (r2 + 2.0 * v2));
    const double drad2 = 2.0 * (k1 + 2.0 * k2 * r2);
    D0 = 1.0 + radial + u2 * drad2 + 2.0 * p1 * vn + 6.0 * p2 * un;
    D1 = uvn * drad2 + 2.0 * p1 * un + 2.0 * p2 * vn;
    D2 = D1;
    D3 = 1.0 + radial + v2 * drad2 + 2.0 * p2 * un + 6.0 * p1 * vn;
    const double dr = r2 * (dk[4] + r2 * dk[5]);  // d radial
    su = un * dr + 2.0 * uvn * dk[6] + (r2 + 2.0 * u2) * dk[7];
    sv = vn * dr + (r2 + 2.0 * v2) * dk[6] + 2.0 * uvn * dk[7];
  }
  u = fx * ud + cam[2];
  v = fy * vd + cam[3];
  double dz0 = 0.0, dz1 = 0.0, dz2 = 1.0;
  if (model == MAVBA_M_CATA) {
    if (nrm > 0.0) {
      const double s = cam[8] / nrm;
      dz0 = s * Xc[0]; dz1 = s * Xc[1]; dz2 = 1.0 + s * Xc[2];
    }
    const double dun = -un * nrm * iz, dvn = -vn * nrm * iz;  // d(un, vn)/d xi
    su += (D0 * dun + D1 * dvn) * dk[8];
    sv += (D2 * dun + D3 * dvn) * dk[8];
  }
  const double n0 = iz - un * iz * dz0, n1 = -un * iz * dz1, n2 = -un * iz * dz2;
  const double n3 = -vn * iz * dz0, n4 = iz - vn * iz * dz1, n5 = -vn * iz * dz2;
  A[0] = fx * (D0 * n0 + D1 * n3); A[1] = fx * (D0 * n1 + D1 * n4); A[2] = fx * (D0 * n2 + D1 * n5);
  A[3] = fy * (D2 * n0 + D3 * n3); A[4] = fy * (D2 * n1 + D3 * n4); A[5] = fy * (D2 * n2 + D3 * n5);
  tk0 = ud * dk[0] + dk[2] + fx * su;
  tk1 = vd * dk[1] + dk[3] + fy * sv;
}

// Raw residual only.
MAVBA_HD void obs_residual(int model, const double* rec, const double* cam, const double* X,
                           double uo, double vo, double* r) {
  double Xr[3], Xc[3], u, v;
  transform_point(rec, X, Xr, Xc);
  project<false>(model, cam, Xc, u, v, nullptr, nullptr);
  r[0] = u - uo; r[1] = v - vo;
}

// Raw residual + Jacobians: Jc[12] = 2x6 (rvec | t), Jp[6] = 2x3, Jk[18] = 2x9.
MAVBA_HD void obs_jacobian(int model, const double* rec, const double* cam, const double* X,
                           double uo, double vo, double* r, double* Jc, double* Jp, double* Jk) {
  double Xr[3], Xc[3], u, v, A[6];
  transform_point(rec, X, Xr, Xc);
  project<true>(model, cam, Xc, u, v, A, Jk);
  r[0] = u - uo; r[1] = v - vo;
  const double a = rec[6], b = rec[7], c = rec[8];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double* Ai = &A[3 * i];
    double m1[3], m2[3];
    // d r/d X = A R: A_i + a (A_i x w) + b ((A_i x w) x w)
    cross3(Ai, rec, m1);
    cross3(m1, rec, m2);
    Jp[3 * i + 0] = Ai[0] + a * m1[0] + b * m2[0];
    Jp[3 * i + 1] = Ai[1] + a * m1[1] + b * m2[1];
    Jp[3 * i + 2] = Ai[2] + a * m1[2] + b * m2[2];
    // d r/d w = -(A_i x Xr) Jl: n + b (n x w) + c ((n x w) x w), n = A_i x Xr
    double n[3];
    cross3(Ai, Xr, n);
    cross3(n, rec, m1);
    cross3(m1, rec, m2);
    Jc[6 * i + 0] = -(n[0] + b * m1[0] + c * m2[0]);
    Jc[6 * i + 1] = -(n[1] + b * m1[1] + c * m2[1]);
    Jc[6 * i + 2] = -(n[2] + b * m1[2] + c * m2[2]);
    Jc[6 * i + 3] = Ai[0]; Jc[6 * i + 4] = Ai[1]; Jc[6 * i + 5] = Ai[2];
  }
}

// The point back-substitution's term of one observation WITHOUT the Jacobian (round 5): it needs
//   t = Jp^T tau,  tau = J_cam delta_cam = Jc (dw, dt) + Jk dk     (2-vector, later scaled by the loss weight squared)
// and with Jc = [ -(A_i x Xr) Jl | A_i ],  Jp = A R  (obs_jacobian above) both collapse to directional derivatives:
//   Jc (dw, dt) = A (dt + (Jl dw) x Xr)         the first-order motion of the camera-frame point, one cross product
//   Jp^T tau    = R^T (A^T tau)                 two cross products
// instead of the six + four cross products and the 12 + 6 Jacobian entries per observation that were built only to be
// contracted again (k_backsub_points_packed spent two thirds of its FP64 instructions there).
// Returns the raw residual in r (for the loss weight) and tau, t UNWEIGHTED: t = Jp^T (Jc dc + Jk dk).
MAVBA_HD void obs_backsub_term(int model, const double* rec, const double* cam, const double* X, double uo, double vo,
                               const double* dc, const double* dk, double* r, double* t) {
  double Xr[3], Xc[3], u, v, A[6], tk0, tk1;
  transform_point(rec, X, Xr, Xc);
  project_dk(model, cam, Xc, dk, u, v, A, tk0, tk1);
  r[0] = u - uo; r[1] = v - vo;
  const double a = rec[6], b = rec[7], c = rec[8];
  // om = Jl dw = dw + b (w x dw) + c (w x (w x dw))
  double m1[3], m2[3], om[3], dX[3];
  cross3(rec, dc, m1);
  cross3(rec, m1, m2);
  om[0] = dc[0] + b * m1[0] + c * m2[0]; om[1] = dc[1] + b * m1[1] + c * m2[1]; om[2] = dc[2] + b * m1[2] + c * m2[2];
  cross3(om, Xr, dX);
  dX[0] += dc[3]; dX[1] += dc[4]; dX[2] += dc[5];
  const double tau0 = A[0] * dX[0] + A[1] * dX[1] + A[2] * dX[2] + tk0;
  const double tau1 = A[3] * dX[0] + A[4] * dX[1] + A[5] * dX[2] + tk1;
  // q = A^T tau, t = R^T q = q - a (w x q) + b (w x (w x q))
  const double q[3] = {A[0] * tau0 + A[3] * tau1, A[1] * tau0 + A[4] * tau1, A[2] * tau0 + A[5] * tau1};
  cross3(rec, q, m1);
  cross3(rec, m1, m2);
  t[0] = q[0] - a * m1[0] + b * m2[0]; t[1] = q[1] - a * m1[1] + b * m2[1]; t[2] = q[2] - a * m1[2] + b * m2[2];
}

// Cauchy robustifier on s = |r|^2: returns the row weight sqrt(rho') and the
// block cost rho/2. inv_b = 1/a^2, b = a^2.
MAVBA_HD void cauchy_weight(double s, double b, double inv_b, double& w, double& half_rho) {
  const double sum = 1.0 + s * inv_b;
  half_rho = 0.5 * b * log_obs(sum);
  w = rsqrt_obs(sum);
}

// ---------------------------------------------------------------------------
// Rotation prior: res = weight * sqrt( sum_k (R[ia[k]] - R0[k])^2 ), R = R(w)
// column-major, ia = {0,3,6,1,4,7,2,6,8} (the reference's index list including
// its (6,7) pair, bundle_adjustment.cc:88-105). jac[3] = d res / d w.
// R0 (column-major) is precomputed by rot_matrix_colmajor(rvec0).
// ---------------------------------------------------------------------------
MAVBA_HD void rot_matrix_colmajor(const double* w, double* R) {
  double a, b, c;
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  rot_coeffs(th2, a, b, c);
  const double x = w[0], y = w[1], z = w[2];
  // [w]x^2 = w w^T - th2 I
  R[0] = 1.0 + b * (x * x - th2); R[3] = -a * z + b * x * y;      R[6] = a * y + b * x * z;
  R[1] = a * z + b * x * y;       R[4] = 1.0 + b * (y * y - th2); R[7] = -a * x + b * y * z;
  R[2] = -a * y + b * x * z;      R[5] = a * x + b * y * z;       R[8] = 1.0 + b * (z * z - th2);
}

MAVBA_HD void rot_prior_eval(const double* w, const double* R0, double weight, double& res,
                             double* jac) {
  const double x = w[0], y = w[1], z = w[2];
  const double th2 = x * x + y * y + z * z;
  double a, b, c;
  rot_coeffs(th2, a, b, c);
  // a1 = (da/dth)/th, b1 = (db/dth)/th
  double a1, b1;
  if (th2 > 1e-6) {
    const double th = sqrt(th2), s = sin(th), co = cos(th);
    a1 = (th * co - s) / (th2 * th);
    b1 = (th * s - 2.0 * (1.0 - co)) / (th2 * th2);
  } else {
    a1 = -1.0 / 3.0 + th2 * (1.0 / 30.0 - th2 / 840.0);
    b1 = -1.0 / 12.0 + th2 * (1.0 / 180.0 - th2 / 6720.0);
  }
  // row-major W = [w]x, W2 = W W
  const double W[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  const double W2[9] = {x * x - th2, x * y, x * z, x * y, y * y - th2, y * z, x * z, y * z, z * z - th2};
  double R[9];  // row-major
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * W[i] + b * W2[i];
  // column-major index k -> row-major (k%3)*3 + k/3
  const int ia[9] = {0, 3, 6, 1, 4, 7, 2, 6, 8};
  double d[9], acc = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int cm = ia[k];
    d[k] = R[(cm % 3) * 3 + cm / 3] - R0[k];
    acc += d[k] * d[k];
  }
  const double nrm = sqrt(acc);
  res = weight * nrm;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    // dR/dw_m = a E_m + b (E_m W + W E_m) + a1 w_m W + b1 w_m W2,  E_m = [e_m]x
    double E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (m == 0) { E[5] = -1; E[7] = 1; }
    if (m == 1) { E[2] = 1; E[6] = -1; }
    if (m == 2) { E[1] = -1; E[3] = 1; }
    double dR[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double ew = 0.0, we = 0.0;
        for (int k = 0; k < 3; ++k) { ew += E[i * 3 + k] * W[k * 3 + j]; we += W[i * 3 + k] * E[k * 3 + j]; }
        dR[i * 3 + j] = a * E[i * 3 + j] + b * (ew + we) + w[m] * (a1 * W[i * 3 + j] + b1 * W2[i * 3 + j]);
      }
    double s = 0.0;
    for (int k = 0; k < 9; ++k) {
      const int cm = ia[k];
      s += d[k] * dR[(cm % 3) * 3 + cm / 3];
    }
    jac[m] = weight * s / nrm;  // nrm == 0 gives inf/nan exactly as the reference's Jet of sqrt(0)
  }
}

// ---------------------------------------------------------------------------
// 3x3 SPD helpers for the point blocks. Symmetric storage order:
//   [0]=xx [1]=xy [2]=xz [3]=yy [4]=yz [5]=zz
// chol3_inv: C = G G^T (G lower). Returns Gi = G^-1 (lower) as
//   [0]=g00 [1]=g10 [2]=g11 [3]=g20 [4]=g21 [5]=g22 ; false if C is not SPD.
// ---------------------------------------------------------------------------
MAVBA_HD bool chol3_inv(const double* C, double* Gi) {
  const double l00 = sqrt(C[0]);
  const double l10 = C[1] / l00, l20 = C[2] / l00;
  const double d1 = C[3] - l10 * l10;
  const double l11 = sqrt(d1);
  const double l21 = (C[4] - l20 * l10) / l11;
  const double d2 = C[5] - l20 * l20 - l21 * l21;
  const double l22 = sqrt(d2);
  const bool ok = (C[0] > 0.0) && (d1 > 0.0) && (d2 > 0.0);
  const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
  Gi[0] = i00;
  Gi[1] = -l10 * i00 * i11;
  Gi[2] = i11;
  Gi[4] = -l21 * i11 * i22;
  Gi[3] = -(l20 * i00 + l21 * Gi[1]) * i22;
  Gi[5] = i22;
  return ok;
}

// y = Gi x   (Gi lower)
MAVBA_HD void gi_mul(const double* Gi, const double* x, double* y) {
  y[0] = Gi[0] * x[0];
  y[1] = Gi[1] * x[0] + Gi[2] * x[1];
  y[2] = Gi[3] * x[0] + Gi[4] * x[1] + Gi[5] * x[2];
}

// y = Gi^T x
MAVBA_HD void git_mul(const double* Gi, const double* x, double* y) {
  y[0] = Gi[0] * x[0] + Gi[1] * x[1] + Gi[3] * x[2];
  y[1] = Gi[2] * x[1] + Gi[4] * x[2];
  y[2] = Gi[5] * x[2];
}

MAVBA_HD double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// Packed index of (r, c), r <= c, in the upper triangle of an n x n symmetric
// matrix stored row by row: (0,0),(0,1)...(0,n-1),(1,1)...
MAVBA_HD int sym_idx(int r, int c, int n) { return r * n - (r * (r - 1)) / 2 + (c - r); }

}  // namespace mavba
#endif  // MAVBA_BA_MATH_H_
