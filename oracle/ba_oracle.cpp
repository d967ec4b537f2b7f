/*
 * ba_oracle.cpp — CPU oracle for the MAVMAP bundle-adjustment hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under mavmap_amd/ or shim/ may include,
 * link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * What it restates (all file:line are relative to /root/reference):
 *   - camera models           src/base3d/camera_models.h:111-130 (PINHOLE),
 *                             :170-193 + :225-242 (OPENCV), :277-302 + :340-357 (CATA),
 *                             image2world :132-145, :195-223, :304-338
 *   - reprojection functor    src/base3d/bundle_adjustment.h:131-159
 *   - rotation-prior functor  src/base3d/bundle_adjustment.cc:72-111 (incl. the
 *                             rotmat[6]-for-rotmat[5] index quirk at :103)
 *   - problem semantics       src/base3d/bundle_adjustment.cc:449-613 (Cauchy loss :477,
 *                             constant blocks :361-385,:545-549, SPARSE_SCHUR LM :554-569,
 *                             point errors :575-598, return value :610)
 *   - ceres::Solve            Ceres-Solver is an UN-VENDORED third-party dependency
 *                             (CMakeLists.txt:11 FIND_PACKAGE(Ceres REQUIRED); README.md:34-38
 *                             "confirmed to work with 1.7.0 / 1.8.0"; no lock file). Its
 *                             published algorithm is restated here from the Ceres 1.8
 *                             sources as recalled in SURVEY.md §3.4: trust_region_minimizer.cc
 *                             (LM loop, termination order), levenberg_marquardt_strategy.cc
 *                             (diagonal clamp, radius update), corrector.cc (rho''<=0 branch),
 *                             loss_function.cc (CauchyLoss), schur_eliminator_impl.h
 *                             (point e-blocks), rotation.h (AngleAxisRotatePoint,
 *                             AngleAxisToRotationMatrix), autodiff via Jets.
 *
 * PARITY PINNING. The projection half (world2image) is pinned against known
 * answers produced by the reference-compiled camera_models.h (SURVEY.md §8(c);
 * tests/golden/world2image_kat.json) and against the properties asserted by the
 * reference's own src/base3d/camera_models_test.cc:16-55. The solver half
 * (everything Ceres does) is **PARITY UNPINNED**: the reference has no test,
 * fixture or golden output for bundle_adjustment()/pose_refinement()
 * (SURVEY.md §4), and Ceres is not installed here, so the LM trajectory is
 * checked only against independent maths (Jets vs analytic vs finite
 * differences, scipy.optimize.least_squares minima, ground-truth recovery).
 *
 * Jacobians come from forward-mode Jets through the templated functors (what
 * the reference does through ceres::AutoDiffCostFunction) or, with
 * jacobian_mode=1, from a hand-derived analytic form; tests require the two to
 * agree to round-off.
 */
#include "../include/mavba.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------
// Forward-mode dual numbers (the role ceres::Jet plays for the reference).
// ---------------------------------------------------------------------------
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  explicit Jet(double x) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Jet(double x, int k) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
};
template <int N> Jet<N> operator+(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; r.a = x.a + y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
template <int N> Jet<N> operator-(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; r.a = x.a - y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
template <int N> Jet<N> operator-(const Jet<N>& x) {
  Jet<N> r; r.a = -x.a; for (int i = 0; i < N; ++i) r.v[i] = -x.v[i]; return r; }
template <int N> Jet<N> operator*(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; r.a = x.a * y.a; for (int i = 0; i < N; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
template <int N> Jet<N> operator/(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; const double inv = 1.0 / y.a; r.a = x.a * inv;
  for (int i = 0; i < N; ++i) r.v[i] = (x.v[i] - r.a * y.v[i]) * inv; return r; }
template <int N> Jet<N> operator+(const Jet<N>& x, double s) { Jet<N> r = x; r.a += s; return r; }
template <int N> Jet<N> operator+(double s, const Jet<N>& x) { Jet<N> r = x; r.a += s; return r; }
template <int N> Jet<N> operator-(const Jet<N>& x, double s) { Jet<N> r = x; r.a -= s; return r; }
template <int N> Jet<N> operator-(double s, const Jet<N>& x) { Jet<N> r = -x; r.a += s; return r; }
template <int N> Jet<N> operator*(const Jet<N>& x, double s) {
  Jet<N> r; r.a = x.a * s; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * s; return r; }
template <int N> Jet<N> operator*(double s, const Jet<N>& x) { return x * s; }
template <int N> Jet<N> operator/(const Jet<N>& x, double s) { return x * (1.0 / s); }
template <int N> Jet<N>& operator+=(Jet<N>& x, const Jet<N>& y) { x = x + y; return x; }
template <int N> Jet<N> sqrt(const Jet<N>& x) {
  Jet<N> r; r.a = std::sqrt(x.a); const double d = 0.5 / r.a;
  for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * d; return r; }
template <int N> Jet<N> sin(const Jet<N>& x) {
  Jet<N> r; r.a = std::sin(x.a); const double c = std::cos(x.a);
  for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * c; return r; }
template <int N> Jet<N> cos(const Jet<N>& x) {
  Jet<N> r; r.a = std::cos(x.a); const double s = -std::sin(x.a);
  for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * s; return r; }
inline double scalar_of(double x) { return x; }
template <int N> double scalar_of(const Jet<N>& x) { return x.a; }
using std::sqrt; using std::sin; using std::cos;

// ---------------------------------------------------------------------------
// ceres/rotation.h (Ceres <= 1.8), restated.
// ---------------------------------------------------------------------------
template <typename T>
void angle_axis_rotate_point(const T aa[3], const T pt[3], T out[3]) {
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (scalar_of(theta2) > 0.0) {
    // Rodrigues: pt cos + (w x pt) sin + w (w . pt)(1 - cos)
    const T theta = sqrt(theta2);
    const T w[3] = {aa[0] / theta, aa[1] / theta, aa[2] / theta};
    const T costheta = cos(theta);
    const T sintheta = sin(theta);
    const T wxp[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2],
                      w[0] * pt[1] - w[1] * pt[0]};
    const T wdp = w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2];
    for (int i = 0; i < 3; ++i)
      out[i] = pt[i] * costheta + wxp[i] * sintheta + w[i] * (1.0 - costheta) * wdp;
  } else {
    // first-order Taylor branch at theta == 0: R pt = pt + aa x pt
    const T wxp[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2],
                      aa[0] * pt[1] - aa[1] * pt[0]};
    for (int i = 0; i < 3; ++i) out[i] = pt[i] + wxp[i];
  }
}

// Column-major R[0..8], as ceres::AngleAxisToRotationMatrix.
template <typename T>
void angle_axis_to_rotation_matrix(const T aa[3], T R[9]) {
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (scalar_of(theta2) > 0.0) {
    const T theta = sqrt(theta2);
    const T wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const T c = cos(theta);
    const T s = sin(theta);
    const T omc = 1.0 - c;
    R[0] = c + wx * wx * omc;
    R[1] = wz * s + wx * wy * omc;
    R[2] = -(wy * s) + wx * wz * omc;
    R[3] = wx * wy * omc - wz * s;
    R[4] = c + wy * wy * omc;
    R[5] = wx * s + wy * wz * omc;
    R[6] = wy * s + wx * wz * omc;
    R[7] = -(wx * s) + wy * wz * omc;
    R[8] = c + wz * wz * omc;
  } else {
    R[0] = T(1.0); R[1] = aa[2];  R[2] = -aa[1];
    R[3] = -aa[2]; R[4] = T(1.0); R[5] = aa[0];
    R[6] = aa[1];  R[7] = -aa[0]; R[8] = T(1.0);
  }
}

// ---------------------------------------------------------------------------
// Camera models (reference src/base3d/camera_models.h).
// ---------------------------------------------------------------------------
template <typename T>
void distortion(const T& u, const T& v, T& du, T& dv, const T* p) {
  // camera_models.h:225-242 (OPENCV) == :340-357 (CATA)
  const T& k1 = p[4]; const T& k2 = p[5]; const T& p1 = p[6]; const T& p2 = p[7];
  const T u2 = u * u;
  const T uv = u * v;
  const T v2 = v * v;
  const T r2 = u2 + v2;
  const T radial = k1 * r2 + k2 * r2 * r2;
  du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2);
  dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2);
}

template <typename T>
void world2image(int model, const T& x, const T& y, const T& z, T& u, T& v, const T* p) {
  if (model == MAVBA_MODEL_PINHOLE) {            // camera_models.h:111-130
    u = x / z; v = y / z;
  } else if (model == MAVBA_MODEL_OPENCV) {      // :170-193
    u = x / z; v = y / z;
    T du, dv; distortion(u, v, du, dv, p);
    u = u + du; v = v + dv;
  } else {                                       // CATA :277-302
    const T zz = z + p[8] * sqrt(x * x + y * y + z * z);
    u = x / zz; v = y / zz;
    T du, dv; distortion(u, v, du, dv, p);
    u = u + du; v = v + dv;
  }
  u = p[0] * u + p[2];
  v = p[1] * v + p[3];
}

void image2world(int model, double u, double v, double& x, double& y, double& z,
                 const double* p) {
  x = (u - p[2]) / p[0];
  y = (v - p[3]) / p[1];
  if (model == MAVBA_MODEL_PINHOLE) { z = 1; return; }   // :132-145
  double xx = x, yy = y, dx, dy;                         // :195-223 / :304-338
  for (int i = 0; i < 10; ++i) { distortion(xx, yy, dx, dy, p); xx = x - dx; yy = y - dy; }
  x = xx; y = yy;
  if (model == MAVBA_MODEL_OPENCV) { z = 1; return; }
  const double xi = p[8];
  if (xi == 1) {
    z = (1 - xx * xx - yy * yy) / 2;
  } else {
    const double r2 = xx * xx + yy * yy;
    z = 1 - xi * (r2 + 1) / (xi + std::sqrt(1 + (1 - xi * xi) * r2));
  }
}

inline int model_num_params(int model) {
  return model == MAVBA_MODEL_PINHOLE ? 4 : model == MAVBA_MODEL_OPENCV ? 8 : 9;
}

// BACostFunction<Model>::operator() — bundle_adjustment.h:131-159.
template <typename T>
void reprojection_residual(int model, const T rvec[3], const T& tx, const T& ty, const T& tz,
                           const T X[3], const T* cam, const double uv[2], T res[2]) {
  T Xc[3];
  angle_axis_rotate_point(rvec, X, Xc);
  Xc[0] = Xc[0] + tx; Xc[1] = Xc[1] + ty; Xc[2] = Xc[2] + tz;
  T u, v;
  world2image(model, Xc[0], Xc[1], Xc[2], u, v, cam);
  res[0] = u - uv[0];
  res[1] = v - uv[1];
}

// BARotationConstraintCostFunction::operator() — bundle_adjustment.cc:72-111.
// R0 is AngleAxisToRotationMatrix(rvec0), column-major (:57-60).
template <typename T>
T rotation_prior_residual(const T rvec[3], const double R0[9], double weight) {
  T R[9];
  angle_axis_to_rotation_matrix(rvec, R);
  // pairs (index into R, index into R0) exactly as the reference lists them;
  // the 8th pair is (6,7) where the transpose pattern would need (5,7).
  static const int ia[9] = {0, 3, 6, 1, 4, 7, 2, 6, 8};
  T acc(0.0);
  for (int k = 0; k < 9; ++k) {
    const T d = R[ia[k]] - R0[k];
    acc = acc + d * d;
  }
  return weight * sqrt(acc);
}

// ---------------------------------------------------------------------------
// Jacobians of one observation. Layout: Jc[2][6] = d r / d(rvec, tx, ty, tz),
// Jp[2][3] = d r / d X, Jk[2][9] = d r / d intrinsics (cols >= K are zero).
// ---------------------------------------------------------------------------
template <int K>
void jet_jacobian(int model, const double* pose, const double* X, const double* cam,
                  const double uv[2], double r[2], double Jc[12], double Jp[6], double Jk[18]) {
  constexpr int N = 9 + K;
  typedef Jet<N> J;
  J rvec[3] = {J(pose[0], 0), J(pose[1], 1), J(pose[2], 2)};
  J tx(pose[3], 3), ty(pose[4], 4), tz(pose[5], 5);
  J Xj[3] = {J(X[0], 6), J(X[1], 7), J(X[2], 8)};
  J camj[9];
  for (int k = 0; k < K; ++k) camj[k] = J(cam[k], 9 + k);
  J res[2];
  reprojection_residual(model, rvec, tx, ty, tz, Xj, camj, uv, res);
  for (int i = 0; i < 2; ++i) {
    r[i] = res[i].a;
    for (int c = 0; c < 6; ++c) Jc[i * 6 + c] = res[i].v[c];
    for (int c = 0; c < 3; ++c) Jp[i * 3 + c] = res[i].v[6 + c];
    for (int c = 0; c < 9; ++c) Jk[i * 9 + c] = c < K ? res[i].v[9 + c] : 0.0;
  }
}

// Hand-derived analytic Jacobian (SURVEY.md §3.4 "Analytic Jacobian").
void analytic_jacobian(int model, const double* pose, const double* X, const double* cam,
                       const double uv[2], double r[2], double Jc[12], double Jp[6],
                       double Jk[18]) {
  const double wx = pose[0], wy = pose[1], wz = pose[2];
  const double th2 = wx * wx + wy * wy + wz * wz;
  // R = I + a [w]x + b [w]x^2 ; left Jacobian Jl = I + b [w]x + c [w]x^2
  double a, b, c;
  if (th2 > 1e-8) {
    const double th = std::sqrt(th2);
    a = std::sin(th) / th;
    b = (1.0 - std::cos(th)) / th2;
    c = (th - std::sin(th)) / (th2 * th);
  } else {
    a = 1.0 - th2 / 6.0 + th2 * th2 / 120.0;
    b = 0.5 - th2 / 24.0 + th2 * th2 / 720.0;
    c = 1.0 / 6.0 - th2 / 120.0 + th2 * th2 / 5040.0;
  }
  const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};  // row-major [w]x
  double W2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0; for (int k = 0; k < 3; ++k) s += W[i * 3 + k] * W[k * 3 + j];
      W2[i * 3 + j] = s;
    }
  double R[9], Jl[9];
  for (int i = 0; i < 9; ++i) {
    const double id = (i % 4 == 0) ? 1.0 : 0.0;
    R[i] = id + a * W[i] + b * W2[i];
    Jl[i] = id + b * W[i] + c * W2[i];
  }
  double Xr[3], Xc[3];
  for (int i = 0; i < 3; ++i) Xr[i] = R[i * 3] * X[0] + R[i * 3 + 1] * X[1] + R[i * 3 + 2] * X[2];
  Xc[0] = Xr[0] + pose[3]; Xc[1] = Xr[1] + pose[4]; Xc[2] = Xr[2] + pose[5];

  const int K = model_num_params(model);
  const double fx = cam[0], fy = cam[1];
  // normalised coordinates and d(un,vn)/dXc
  double un, vn, dn[6];  // dn = [dun/dx dun/dy dun/dz ; dvn/dx dvn/dy dvn/dz]
  double nrm = 0, zz = Xc[2];
  if (model == MAVBA_MODEL_CATA) {
    nrm = std::sqrt(Xc[0] * Xc[0] + Xc[1] * Xc[1] + Xc[2] * Xc[2]);
    zz = Xc[2] + cam[8] * nrm;
  }
  const double iz = 1.0 / zz;
  un = Xc[0] * iz; vn = Xc[1] * iz;
  double dzz[3] = {0, 0, 1};
  if (model == MAVBA_MODEL_CATA && nrm > 0) {
    dzz[0] = cam[8] * Xc[0] / nrm; dzz[1] = cam[8] * Xc[1] / nrm; dzz[2] = 1 + cam[8] * Xc[2] / nrm;
  }
  dn[0] = iz - un * iz * dzz[0]; dn[1] = -un * iz * dzz[1]; dn[2] = -un * iz * dzz[2];
  dn[3] = -vn * iz * dzz[0]; dn[4] = iz - vn * iz * dzz[1]; dn[5] = -vn * iz * dzz[2];

  double ud = un, vd = vn;            // distorted normalised coords
  double D[4] = {1, 0, 0, 1};         // d(ud,vd)/d(un,vn)
  for (int i = 0; i < 18; ++i) Jk[i] = 0.0;
  if (model != MAVBA_MODEL_PINHOLE) {
    const double k1 = cam[4], k2 = cam[5], p1 = cam[6], p2 = cam[7];
    const double u2 = un * un, v2 = vn * vn, uvn = un * vn, r2 = u2 + v2;
    const double radial = k1 * r2 + k2 * r2 * r2;
    const double drad = k1 + 2 * k2 * r2;  // d radial / d r2
    ud = un + un * radial + 2 * p1 * uvn + p2 * (r2 + 2 * u2);
    vd = vn + vn * radial + 2 * p2 * uvn + p1 * (r2 + 2 * v2);
    D[0] = 1 + radial + un * drad * 2 * un + 2 * p1 * vn + p2 * (2 * un + 4 * un);
    D[1] = un * drad * 2 * vn + 2 * p1 * un + p2 * 2 * vn;
    D[2] = vn * drad * 2 * un + 2 * p2 * vn + p1 * 2 * un;
    D[3] = 1 + radial + vn * drad * 2 * vn + 2 * p2 * un + p1 * (2 * vn + 4 * vn);
    // d r / d (k1,k2,p1,p2)
    Jk[4] = fx * un * r2;      Jk[9 + 4] = fy * vn * r2;
    Jk[5] = fx * un * r2 * r2; Jk[9 + 5] = fy * vn * r2 * r2;
    Jk[6] = fx * 2 * uvn;      Jk[9 + 6] = fy * (r2 + 2 * v2);
    Jk[7] = fx * (r2 + 2 * u2); Jk[9 + 7] = fy * 2 * uvn;
    if (model == MAVBA_MODEL_CATA) {
      // d(un,vn)/dxi = -(un,vn) * |Xc| / zz, then through the distortion
      const double dun = -un * nrm * iz, dvn = -vn * nrm * iz;
      Jk[8] = fx * (D[0] * dun + D[1] * dvn);
      Jk[9 + 8] = fy * (D[2] * dun + D[3] * dvn);
    }
  }
  Jk[0] = ud; Jk[2] = 1.0; Jk[9 + 1] = vd; Jk[9 + 3] = 1.0;
  r[0] = fx * ud + cam[2] - uv[0];
  r[1] = fy * vd + cam[3] - uv[1];
  (void)K;

  // dC/dXc (2x3)
  double A[6];
  for (int j = 0; j < 3; ++j) {
    A[j] = fx * (D[0] * dn[j] + D[1] * dn[3 + j]);
    A[3 + j] = fy * (D[2] * dn[j] + D[3] * dn[3 + j]);
  }
  // d r / d t = A ; d r / d X = A R ; d r / d w = -A [Xr]x Jl
  const double Xx[9] = {0, -Xr[2], Xr[1], Xr[2], 0, -Xr[0], -Xr[1], Xr[0], 0};
  for (int i = 0; i < 2; ++i) {
    double AX[3];
    for (int j = 0; j < 3; ++j) {
      double s = 0, p = 0;
      for (int k = 0; k < 3; ++k) { s += A[i * 3 + k] * R[k * 3 + j]; p += A[i * 3 + k] * Xx[k * 3 + j]; }
      Jp[i * 3 + j] = s; AX[j] = p;
    }
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += AX[k] * Jl[k * 3 + j];
      Jc[i * 6 + j] = -s;
      Jc[i * 6 + 3 + j] = A[i * 3 + j];
    }
  }
}

void obs_jacobian(int mode, int model, const double* pose, const double* X, const double* cam,
                  const double uv[2], double r[2], double Jc[12], double Jp[6], double Jk[18]) {
  if (mode == 1) { analytic_jacobian(model, pose, X, cam, uv, r, Jc, Jp, Jk); return; }
  if (model == MAVBA_MODEL_PINHOLE) jet_jacobian<4>(model, pose, X, cam, uv, r, Jc, Jp, Jk);
  else if (model == MAVBA_MODEL_OPENCV) jet_jacobian<8>(model, pose, X, cam, uv, r, Jc, Jp, Jk);
  else jet_jacobian<9>(model, pose, X, cam, uv, r, Jc, Jp, Jk);
}

void obs_residual(int model, const double* pose, const double* X, const double* cam,
                  const double uv[2], double r[2]) {
  reprojection_residual<double>(model, pose, pose[3], pose[4], pose[5], X, cam, uv, r);
}

// ceres::CauchyLoss::Evaluate (loss_function.cc): b = a^2, c = 1/b.
inline void cauchy(double a, double s, double rho[3]) {
  const double b = a * a, c = 1.0 / b;
  const double sum = 1.0 + s * c;
  const double inv = 1.0 / sum;
  rho[0] = b * std::log(sum);
  rho[1] = inv;
  rho[2] = -c * (inv * inv);
}

// ---------------------------------------------------------------------------
// The reduced program (ceres SolverImpl::CreateReducedProgram semantics).
// ---------------------------------------------------------------------------
struct Program {
  const mavba_problem* P;
  mavba_options opt;
  int jac_mode;
  int NI, NC, NP;
  int64_t NO;
  std::vector<int> K;               // params per camera
  // column maps into the reduced camera system (-1 = constant / absent)
  std::vector<int> col_pose;        // [NI*6]
  std::vector<int> col_intr;        // [NC]  start column of the K-block
  std::vector<int> idx_point;       // [NP]  index among free points or -1
  int n_cam;                        // reduced camera-system dimension
  int n_fp;                         // number of free (eliminated) points
  std::vector<int64_t> kept;        // kept observation ids, caller order
  std::vector<int> kept_prior;      // kept rotation priors
  std::vector<double> prior_R0;     // [num_rot_priors][9]
  double fixed_cost;
  int64_t num_residuals, num_residuals_reduced, num_parameters_reduced;
  // point-major view of kept observations (for the Schur eliminator)
  std::vector<int64_t> pt_start;    // [NP+1] into pt_obs
  std::vector<int64_t> pt_obs;      // indices into `kept`
  // working state
  std::vector<double> poses, intr, points;
  // image-major view + row envelope (linear-solver mode 1 only; built on first use)
  mutable std::vector<int64_t> img_start, img_obs;
  mutable std::vector<int> env_first;
};

inline bool pose_col_const(const mavba_problem* P, int img, int c) {
  const unsigned m = P->pose_const ? P->pose_const[img] : 0u;
  if (c < 3) return (m & MAVBA_CONST_RVEC) != 0;
  return (m & (MAVBA_CONST_TX << (c - 3))) != 0;
}

int build_program(Program& G, const mavba_problem* P, const mavba_options* opt, int jac_mode) {
  G.P = P; G.opt = *opt; G.jac_mode = jac_mode;
  G.NI = P->num_images; G.NC = P->num_cameras; G.NP = P->num_points; G.NO = P->num_obs;
  G.K.resize(G.NC);
  for (int c = 0; c < G.NC; ++c) {
    const int m = P->camera_model[c];
    if (m < 1 || m > 3) return MAVBA_ERR_BAD_MODEL;
    G.K[c] = model_num_params(m);
  }
  for (int64_t o = 0; o < G.NO; ++o) {
    if (P->obs_image[o] < 0 || P->obs_image[o] >= G.NI || P->obs_point[o] < 0 ||
        P->obs_point[o] >= G.NP) return MAVBA_ERR_BAD_INDEX;
  }
  for (int i = 0; i < G.NI; ++i)
    if (P->image_camera[i] < 0 || P->image_camera[i] >= G.NC) return MAVBA_ERR_BAD_INDEX;
  G.poses.assign(P->poses, P->poses + (size_t)G.NI * 6);
  G.intr.assign(P->intrinsics, P->intrinsics + (size_t)G.NC * MAVBA_MAX_INTR);
  G.points.assign(P->points, P->points + (size_t)G.NP * 3);

  // Which blocks are constant.
  std::vector<char> img_all_const(G.NI), cam_const(G.NC), pt_const(G.NP);
  for (int i = 0; i < G.NI; ++i) img_all_const[i] = ((P->pose_const ? P->pose_const[i] : 0) & 15u) == 15u;
  for (int c = 0; c < G.NC; ++c) cam_const[c] = P->intr_const ? P->intr_const[c] != 0 : 0;
  for (int p = 0; p < G.NP; ++p) pt_const[p] = P->point_const ? P->point_const[p] != 0 : 0;

  // Residual blocks whose parameter blocks are all constant leave the program;
  // their cost is the fixed cost (ceres RemoveFixedBlocksFromProgram).
  std::vector<char> img_used(G.NI, 0), cam_used(G.NC, 0), pt_used(G.NP, 0);
  G.kept.clear(); G.fixed_cost = 0.0;
  for (int64_t o = 0; o < G.NO; ++o) {
    const int i = P->obs_image[o], p = P->obs_point[o], c = P->image_camera[i];
    if (img_all_const[i] && cam_const[c] && pt_const[p]) {
      double r[2];
      obs_residual(P->camera_model[c], &G.poses[i * 6], &G.points[p * 3], &G.intr[c * 9],
                   &P->obs_uv[o * 2], r);
      double rho[3]; cauchy(opt->loss_scale_factor, r[0] * r[0] + r[1] * r[1], rho);
      G.fixed_cost += 0.5 * rho[0];
      continue;
    }
    G.kept.push_back(o);
    img_used[i] = 1; cam_used[c] = 1; pt_used[p] = 1;
  }
  G.prior_R0.assign((size_t)P->num_rot_priors * 9, 0.0);
  G.kept_prior.clear();
  for (int q = 0; q < P->num_rot_priors; ++q) {
    const int i = P->rot_prior_image[q];
    if (i < 0 || i >= G.NI) return MAVBA_ERR_BAD_INDEX;
    angle_axis_to_rotation_matrix<double>(&P->rot_prior_rvec[q * 3], &G.prior_R0[q * 9]);
    if (pose_col_const(P, i, 0)) {
      const double r = rotation_prior_residual<double>(&G.poses[i * 6], &G.prior_R0[q * 9],
                                                       P->rot_prior_weight);
      G.fixed_cost += 0.5 * r * r;  // NULL loss
      continue;
    }
    G.kept_prior.push_back(q);
    img_used[i] = 1;
  }
  G.num_residuals = 2 * G.NO + P->num_rot_priors;
  G.num_residuals_reduced = 2 * (int64_t)G.kept.size() + (int64_t)G.kept_prior.size();

  // Column maps.
  G.col_pose.assign((size_t)G.NI * 6, -1);
  G.col_intr.assign(G.NC, -1);
  G.idx_point.assign(G.NP, -1);
  int n = 0;
  for (int i = 0; i < G.NI; ++i) {
    if (!img_used[i]) continue;
    for (int c = 0; c < 6; ++c)
      if (!pose_col_const(P, i, c)) G.col_pose[i * 6 + c] = n++;
  }
  for (int c = 0; c < G.NC; ++c)
    if (cam_used[c] && !cam_const[c]) { G.col_intr[c] = n; n += G.K[c]; }
  G.n_cam = n;
  int nf = 0;
  for (int p = 0; p < G.NP; ++p)
    if (pt_used[p] && !pt_const[p]) G.idx_point[p] = nf++;
  G.n_fp = nf;
  G.num_parameters_reduced = (int64_t)n + 3 * (int64_t)nf;

  // point-major view
  G.pt_start.assign((size_t)G.NP + 1, 0);
  for (size_t k = 0; k < G.kept.size(); ++k) G.pt_start[P->obs_point[G.kept[k]] + 1]++;
  for (int p = 0; p < G.NP; ++p) G.pt_start[p + 1] += G.pt_start[p];
  G.pt_obs.resize(G.kept.size());
  std::vector<int64_t> cur(G.pt_start.begin(), G.pt_start.end() - 1);
  for (size_t k = 0; k < G.kept.size(); ++k) G.pt_obs[cur[P->obs_point[G.kept[k]]]++] = (int64_t)k;
  return MAVBA_OK;
}

extern int g_linear_solver_mode;
struct Lin;
void accumulate_columns_parallel(const Program& G, const Lin& L, bool squares, std::vector<double>& out);

// Linearisation at a point: loss-corrected residuals and Jacobian blocks.
struct Lin {
  std::vector<double> r;    // [nk][2]
  std::vector<double> Jc;   // [nk][2][6]
  std::vector<double> Jp;   // [nk][2][3]
  std::vector<double> Jk;   // [nk][2][9]
  std::vector<double> pr;   // [nprior]
  std::vector<double> pJ;   // [nprior][3]
  std::vector<double> grad; // [n_cam + 3 n_fp]
  double cost;
};

// Evaluate at (poses, intr, points). jac=false: cost only.
void evaluate(const Program& G, const double* poses, const double* intr, const double* points,
              bool jac, Lin& L, double* cost_out) {
  const mavba_problem* P = G.P;
  const size_t nk = G.kept.size();
  if (jac) {
    L.r.resize(nk * 2); L.Jc.resize(nk * 12); L.Jp.resize(nk * 6); L.Jk.resize(nk * 18);
    L.pr.resize(G.kept_prior.size()); L.pJ.resize(G.kept_prior.size() * 3);
  }
  double cost = 0.0;
  const double a = G.opt.loss_scale_factor;
#pragma omp parallel for schedule(static) reduction(+ : cost)
  for (int64_t k = 0; k < (int64_t)nk; ++k) {
    const int64_t o = G.kept[k];
    const int i = P->obs_image[o], p = P->obs_point[o], c = P->image_camera[i];
    double r[2], Jc[12], Jp[6], Jk[18];
    if (jac) obs_jacobian(G.jac_mode, P->camera_model[c], &poses[i * 6], &points[p * 3],
                          &intr[c * 9], &P->obs_uv[o * 2], r, Jc, Jp, Jk);
    else obs_residual(P->camera_model[c], &poses[i * 6], &points[p * 3], &intr[c * 9],
                      &P->obs_uv[o * 2], r);
    double rho[3]; cauchy(a, r[0] * r[0] + r[1] * r[1], rho);
    cost += 0.5 * rho[0];
    if (jac) {
      // ceres Corrector with rho'' <= 0 (always true for Cauchy): scale the
      // residual and every Jacobian row by sqrt(rho').
      const double w = std::sqrt(rho[1]);
      L.r[k * 2] = w * r[0]; L.r[k * 2 + 1] = w * r[1];
      for (int e = 0; e < 12; ++e) L.Jc[k * 12 + e] = w * Jc[e];
      for (int e = 0; e < 6; ++e) L.Jp[k * 6 + e] = w * Jp[e];
      for (int e = 0; e < 18; ++e) L.Jk[k * 18 + e] = w * Jk[e];
    }
  }
  for (size_t q = 0; q < G.kept_prior.size(); ++q) {
    const int pq = G.kept_prior[q];
    const int i = P->rot_prior_image[pq];
    if (jac) {
      typedef Jet<3> J;
      J rv[3] = {J(poses[i * 6], 0), J(poses[i * 6 + 1], 1), J(poses[i * 6 + 2], 2)};
      const J res = rotation_prior_residual<J>(rv, &G.prior_R0[pq * 9], P->rot_prior_weight);
      L.pr[q] = res.a;
      for (int e = 0; e < 3; ++e) L.pJ[q * 3 + e] = res.v[e];
      cost += 0.5 * res.a * res.a;
    } else {
      const double res = rotation_prior_residual<double>(&poses[i * 6], &G.prior_R0[pq * 9],
                                                         P->rot_prior_weight);
      cost += 0.5 * res * res;
    }
  }
  if (jac && g_linear_solver_mode == 1) {
    accumulate_columns_parallel(G, L, false, L.grad);
    L.cost = cost;
  } else if (jac) {
    // gradient = J^T r over the reduced parameter vector [cameras | points]
    L.grad.assign((size_t)G.n_cam + 3 * (size_t)G.n_fp, 0.0);
    for (size_t k = 0; k < nk; ++k) {
      const int64_t o = G.kept[k];
      const int i = P->obs_image[o], p = P->obs_point[o], c = P->image_camera[i];
      for (int row = 0; row < 2; ++row) {
        const double rr = L.r[k * 2 + row];
        for (int e = 0; e < 6; ++e) {
          const int col = G.col_pose[i * 6 + e];
          if (col >= 0) L.grad[col] += L.Jc[k * 12 + row * 6 + e] * rr;
        }
        if (G.col_intr[c] >= 0)
          for (int e = 0; e < G.K[c]; ++e) L.grad[G.col_intr[c] + e] += L.Jk[k * 18 + row * 9 + e] * rr;
        if (G.idx_point[p] >= 0)
          for (int e = 0; e < 3; ++e)
            L.grad[G.n_cam + 3 * G.idx_point[p] + e] += L.Jp[k * 6 + row * 3 + e] * rr;
      }
    }
    for (size_t q = 0; q < G.kept_prior.size(); ++q) {
      const int i = P->rot_prior_image[G.kept_prior[q]];
      for (int e = 0; e < 3; ++e) {
        const int col = G.col_pose[i * 6 + e];
        if (col >= 0) L.grad[col] += L.pJ[q * 3 + e] * L.pr[q];
      }
    }
    L.cost = cost;
  }
  if (cost_out) *cost_out = cost;
}

// Squared column norms of the (current) Jacobian over the reduced parameters.
void column_sq_norms(const Program& G, const Lin& L, std::vector<double>& out) {
  if (g_linear_solver_mode == 1) { accumulate_columns_parallel(G, L, true, out); return; }
  const mavba_problem* P = G.P;
  out.assign((size_t)G.n_cam + 3 * (size_t)G.n_fp, 0.0);
  for (size_t k = 0; k < G.kept.size(); ++k) {
    const int64_t o = G.kept[k];
    const int i = P->obs_image[o], p = P->obs_point[o], c = P->image_camera[i];
    for (int row = 0; row < 2; ++row) {
      for (int e = 0; e < 6; ++e) {
        const int col = G.col_pose[i * 6 + e];
        if (col >= 0) { const double x = L.Jc[k * 12 + row * 6 + e]; out[col] += x * x; }
      }
      if (G.col_intr[c] >= 0)
        for (int e = 0; e < G.K[c]; ++e) { const double x = L.Jk[k * 18 + row * 9 + e]; out[G.col_intr[c] + e] += x * x; }
      if (G.idx_point[p] >= 0)
        for (int e = 0; e < 3; ++e) { const double x = L.Jp[k * 6 + row * 3 + e]; out[G.n_cam + 3 * G.idx_point[p] + e] += x * x; }
    }
  }
  for (size_t q = 0; q < G.kept_prior.size(); ++q) {
    const int i = P->rot_prior_image[G.kept_prior[q]];
    for (int e = 0; e < 3; ++e) {
      const int col = G.col_pose[i * 6 + e];
      if (col >= 0) out[col] += L.pJ[q * 3 + e] * L.pJ[q * 3 + e];
    }
  }
}

// jacobian->ScaleColumns(scale)
void scale_columns(const Program& G, Lin& L, const std::vector<double>& scale) {
  const mavba_problem* P = G.P;
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < (int64_t)G.kept.size(); ++k) {
    const int64_t o = G.kept[k];
    const int i = P->obs_image[o], p = P->obs_point[o], c = P->image_camera[i];
    for (int row = 0; row < 2; ++row) {
      for (int e = 0; e < 6; ++e) {
        const int col = G.col_pose[i * 6 + e];
        if (col >= 0) L.Jc[k * 12 + row * 6 + e] *= scale[col];
      }
      if (G.col_intr[c] >= 0)
        for (int e = 0; e < G.K[c]; ++e) L.Jk[k * 18 + row * 9 + e] *= scale[G.col_intr[c] + e];
      if (G.idx_point[p] >= 0)
        for (int e = 0; e < 3; ++e) L.Jp[k * 6 + row * 3 + e] *= scale[G.n_cam + 3 * G.idx_point[p] + e];
    }
  }
  for (size_t q = 0; q < G.kept_prior.size(); ++q) {
    const int i = P->rot_prior_image[G.kept_prior[q]];
    for (int e = 0; e < 3; ++e) {
      const int col = G.col_pose[i * 6 + e];
      if (col >= 0) L.pJ[q * 3 + e] *= scale[col];
    }
  }
}

// Dense Cholesky (lower, in place, row-major). Returns false if not SPD.
bool dense_cholesky(int n, double* A) {
  const int NB = 64;
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int kb = std::min(NB, n - k0);
    // factor the diagonal block
    for (int j = k0; j < k0 + kb; ++j) {
      double d = A[(size_t)j * n + j];
      for (int k = k0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
      if (!(d > 0.0) || !std::isfinite(d)) return false;
      d = std::sqrt(d);
      A[(size_t)j * n + j] = d;
      for (int i = j + 1; i < k0 + kb; ++i) {
        double s = A[(size_t)i * n + j];
        for (int k = k0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
        A[(size_t)i * n + j] = s / d;
      }
    }
    const int r0 = k0 + kb;
    // panel below: A[i, k0:k0+kb] <- A[i, .] L_kk^-T
#pragma omp parallel for schedule(static)
    for (int i = r0; i < n; ++i) {
      for (int j = k0; j < k0 + kb; ++j) {
        double s = A[(size_t)i * n + j];
        for (int k = k0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
        A[(size_t)i * n + j] = s / A[(size_t)j * n + j];
      }
    }
    // trailing update (lower triangle only)
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = r0; i < n; ++i) {
      const double* ai = &A[(size_t)i * n + k0];
      for (int j = r0; j <= i; ++j) {
        const double* aj = &A[(size_t)j * n + k0];
        double s = 0.0;
        for (int k = 0; k < kb; ++k) s += ai[k] * aj[k];
        A[(size_t)i * n + j] -= s;
      }
    }
  }
  return true;
}

void cholesky_solve(int n, const double* L, double* b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * b[k];
    b[i] = s / L[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * b[k];
    b[i] = s / L[(size_t)i * n + i];
  }
}

inline bool invert3(const double A[9], double inv[9]) {
  const double c0 = A[4] * A[8] - A[5] * A[7];
  const double c1 = A[5] * A[6] - A[3] * A[8];
  const double c2 = A[3] * A[7] - A[4] * A[6];
  const double det = A[0] * c0 + A[1] * c1 + A[2] * c2;
  if (det == 0.0 || !std::isfinite(det)) return false;
  const double id = 1.0 / det;
  inv[0] = c0 * id; inv[1] = (A[2] * A[7] - A[1] * A[8]) * id; inv[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  inv[3] = c1 * id; inv[4] = (A[0] * A[8] - A[2] * A[6]) * id; inv[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  inv[6] = c2 * id; inv[7] = (A[1] * A[6] - A[0] * A[7]) * id; inv[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  return true;
}

// Camera-side row of one kept observation: up to 6 + K (column, value) pairs.
inline int camera_row(const Program& G, const Lin& L, size_t k, int row, int cols[15], double vals[15]) {
  const mavba_problem* P = G.P;
  const int64_t o = G.kept[k];
  const int i = P->obs_image[o], c = P->image_camera[i];
  int m = 0;
  for (int e = 0; e < 6; ++e) {
    const int col = G.col_pose[i * 6 + e];
    if (col >= 0) { cols[m] = col; vals[m++] = L.Jc[k * 12 + row * 6 + e]; }
  }
  if (G.col_intr[c] >= 0)
    for (int e = 0; e < G.K[c]; ++e) { cols[m] = G.col_intr[c] + e; vals[m++] = L.Jk[k * 18 + row * 9 + e]; }
  return m;
}

// SchurComplementSolver: eliminate the free points, form S and v (dense).
// D is the LM diagonal over [cameras | points]. Returns per-point inverse
// blocks and g_p for the back-substitution.
void schur_eliminate(const Program& G, const Lin& L, const std::vector<double>& D,
                     std::vector<double>& S, std::vector<double>& v,
                     std::vector<double>& ete_inv, std::vector<double>& gp) {
  const mavba_problem* P = G.P;
  const int n = G.n_cam;
  S.assign((size_t)n * n, 0.0);
  v.assign(n, 0.0);
  ete_inv.assign((size_t)G.n_fp * 9, 0.0);
  gp.assign((size_t)G.n_fp * 3, 0.0);
  for (int j = 0; j < n; ++j) S[(size_t)j * n + j] = D[j] * D[j];
  // rotation-prior rows: f-blocks only
  for (size_t q = 0; q < G.kept_prior.size(); ++q) {
    const int i = P->rot_prior_image[G.kept_prior[q]];
    for (int a = 0; a < 3; ++a) {
      const int ca = G.col_pose[i * 6 + a];
      if (ca < 0) continue;
      v[ca] += L.pJ[q * 3 + a] * L.pr[q];
      for (int b = 0; b < 3; ++b) {
        const int cb = G.col_pose[i * 6 + b];
        if (cb >= 0) S[(size_t)ca * n + cb] += L.pJ[q * 3 + a] * L.pJ[q * 3 + b];
      }
    }
  }
  std::vector<int> rc;      // columns touched by a point
  std::vector<double> W;    // (F^T E) rows for those columns, [ncols][3]
  std::vector<int> colpos(n, -1);
  for (int p = 0; p < G.NP; ++p) {
    const int64_t b0 = G.pt_start[p], b1 = G.pt_start[p + 1];
    if (b0 == b1) continue;
    const int fp = G.idx_point[p];
    double ete[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    rc.clear(); W.clear();
    for (int64_t t = b0; t < b1; ++t) {
      const size_t k = (size_t)G.pt_obs[t];
      for (int row = 0; row < 2; ++row) {
        int cols[15]; double vals[15];
        const int m = camera_row(G, L, k, row, cols, vals);
        const double rr = L.r[k * 2 + row];
        const double* e = &L.Jp[k * 6 + row * 3];
        // F^T F and F^T r
        for (int x = 0; x < m; ++x) {
          v[cols[x]] += vals[x] * rr;
          for (int y = 0; y < m; ++y) S[(size_t)cols[x] * n + cols[y]] += vals[x] * vals[y];
        }
        if (fp < 0) continue;
        for (int x = 0; x < 3; ++x) {
          g[x] += e[x] * rr;
          for (int y = 0; y < 3; ++y) ete[x * 3 + y] += e[x] * e[y];
        }
        for (int x = 0; x < m; ++x) {
          int pos = colpos[cols[x]];
          if (pos < 0) { pos = (int)rc.size(); colpos[cols[x]] = pos; rc.push_back(cols[x]); W.insert(W.end(), 3, 0.0); }
          for (int y = 0; y < 3; ++y) W[pos * 3 + y] += vals[x] * e[y];
        }
      }
    }
    if (fp < 0) continue;
    for (int x = 0; x < 3; ++x) { const double d = D[n + 3 * fp + x]; ete[x * 4] += d * d; }
    double inv[9];
    if (!invert3(ete, inv)) { for (int x = 0; x < 9; ++x) inv[x] = std::numeric_limits<double>::quiet_NaN(); }
    for (int x = 0; x < 9; ++x) ete_inv[(size_t)fp * 9 + x] = inv[x];
    for (int x = 0; x < 3; ++x) gp[(size_t)fp * 3 + x] = g[x];
    const int m = (int)rc.size();
    // S -= W inv W^T ; v -= W inv g
    double ig[3];
    for (int x = 0; x < 3; ++x) ig[x] = inv[x * 3] * g[0] + inv[x * 3 + 1] * g[1] + inv[x * 3 + 2] * g[2];
    for (int x = 0; x < m; ++x) {
      double wi[3];
      for (int y = 0; y < 3; ++y)
        wi[y] = W[x * 3] * inv[y] + W[x * 3 + 1] * inv[3 + y] + W[x * 3 + 2] * inv[6 + y];
      v[rc[x]] -= W[x * 3] * ig[0] + W[x * 3 + 1] * ig[1] + W[x * 3 + 2] * ig[2];
      double* Srow = &S[(size_t)rc[x] * n];
      for (int y = 0; y < m; ++y)
        Srow[rc[y]] -= wi[0] * W[y * 3] + wi[1] * W[y * 3 + 1] + wi[2] * W[y * 3 + 2];
    }
    for (int x = 0; x < m; ++x) colpos[rc[x]] = -1;
  }
}

// y_p = ete^-1 (g_p - E^T F y_c)
void back_substitute(const Program& G, const Lin& L, const std::vector<double>& ete_inv,
                     const std::vector<double>& gp, const double* yc, double* yp) {
#pragma omp parallel for schedule(static)
  for (int p = 0; p < G.NP; ++p) {
    const int fp = G.idx_point[p];
    if (fp < 0) continue;
    double t[3] = {gp[(size_t)fp * 3], gp[(size_t)fp * 3 + 1], gp[(size_t)fp * 3 + 2]};
    for (int64_t s = G.pt_start[p]; s < G.pt_start[p + 1]; ++s) {
      const size_t k = (size_t)G.pt_obs[s];
      for (int row = 0; row < 2; ++row) {
        int cols[15]; double vals[15];
        const int m = camera_row(G, L, k, row, cols, vals);
        double fy = 0.0;
        for (int x = 0; x < m; ++x) fy += vals[x] * yc[cols[x]];
        for (int x = 0; x < 3; ++x) t[x] -= L.Jp[k * 6 + row * 3 + x] * fy;
      }
    }
    const double* inv = &ete_inv[(size_t)fp * 9];
    for (int x = 0; x < 3; ++x) yp[3 * fp + x] = inv[x * 3] * t[0] + inv[x * 3 + 1] * t[1] + inv[x * 3 + 2] * t[2];
  }
}

// evaluator->Plus for the reduced parameters.
void plus(const Program& G, const std::vector<double>& poses, const std::vector<double>& intr,
          const std::vector<double>& points, const double* delta, std::vector<double>& poses2,
          std::vector<double>& intr2, std::vector<double>& points2) {
  poses2 = poses; intr2 = intr; points2 = points;
  for (int i = 0; i < G.NI; ++i)
    for (int e = 0; e < 6; ++e) { const int col = G.col_pose[i * 6 + e]; if (col >= 0) poses2[i * 6 + e] += delta[col]; }
  for (int c = 0; c < G.NC; ++c)
    if (G.col_intr[c] >= 0) for (int e = 0; e < G.K[c]; ++e) intr2[c * 9 + e] += delta[G.col_intr[c] + e];
  for (int p = 0; p < G.NP; ++p)
    if (G.idx_point[p] >= 0) for (int e = 0; e < 3; ++e) points2[p * 3 + e] += delta[G.n_cam + 3 * G.idx_point[p] + e];
}

double reduced_norm(const Program& G, const std::vector<double>& poses, const std::vector<double>& intr,
                    const std::vector<double>& points) {
  double s = 0.0;
  for (int i = 0; i < G.NI; ++i)
    for (int e = 0; e < 6; ++e) if (G.col_pose[i * 6 + e] >= 0) s += poses[i * 6 + e] * poses[i * 6 + e];
  for (int c = 0; c < G.NC; ++c)
    if (G.col_intr[c] >= 0) for (int e = 0; e < G.K[c]; ++e) s += intr[c * 9 + e] * intr[c * 9 + e];
  for (int p = 0; p < G.NP; ++p)
    if (G.idx_point[p] >= 0) for (int e = 0; e < 3; ++e) s += points[p * 3 + e] * points[p * 3 + e];
  return std::sqrt(s);
}

// ---------------------------------------------------------------------------
// Linear-solver mode 1: block-sparse Schur complement + envelope (profile) Cholesky, OpenMP.
// The same arithmetic as schur_eliminate / dense_cholesky above, organised the way a sparse CPU solver
// (ceres SPARSE_SCHUR) organises it: S is only touched inside its block structure, every row block has ONE
// owner thread (no atomics: the sums do not depend on the thread count), and the factorisation skips
// everything left of the row envelope. Used for the cpu_baseline of bench.py and for oracle steps at
// sizes where the dense path would take minutes (C5: n = 12 018).
// ---------------------------------------------------------------------------
int g_linear_solver_mode = 0;  // 0 = dense (default), 1 = sparse (declared above)

struct SparseIndex {
  std::vector<int64_t>& img_start;  // [NI+1] into img_obs
  std::vector<int64_t>& img_obs;    // indices into `kept`, grouped by image, kept order inside
  std::vector<int>& first;          // [n_cam] first structurally non-zero column of every row of S
  explicit SparseIndex(const Program& G) : img_start(G.img_start), img_obs(G.img_obs), first(G.env_first) {}
};

void build_sparse_index(const Program& G, SparseIndex& X) {
  const mavba_problem* P = G.P;
  const size_t nk = G.kept.size();
  if (!X.img_start.empty()) return;  // built already
  X.img_start.assign((size_t)G.NI + 1, 0);
  for (size_t k = 0; k < nk; ++k) X.img_start[P->obs_image[G.kept[k]] + 1]++;
  for (int i = 0; i < G.NI; ++i) X.img_start[i + 1] += X.img_start[i];
  X.img_obs.resize(nk);
  std::vector<int64_t> cur(X.img_start.begin(), X.img_start.end() - 1);
  for (size_t k = 0; k < nk; ++k) X.img_obs[cur[P->obs_image[G.kept[k]]]++] = (int64_t)k;
  // first column of an image's pose rows: the lowest pose column among the images that share a point with it
  std::vector<int> img_col(G.NI, -1);
  for (int i = 0; i < G.NI; ++i)
    for (int e = 0; e < 6; ++e) if (G.col_pose[i * 6 + e] >= 0) { img_col[i] = G.col_pose[i * 6 + e]; break; }
  X.first.assign(G.n_cam, 0);
  std::vector<int> img_first(G.NI, 0);
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < G.NI; ++i) {
    int f = img_col[i] < 0 ? 0 : img_col[i];
    for (int64_t t = X.img_start[i]; t < X.img_start[i + 1]; ++t) {
      const int p = P->obs_point[G.kept[X.img_obs[t]]];
      if (G.idx_point[p] < 0) continue;
      for (int64_t u = G.pt_start[p]; u < G.pt_start[p + 1]; ++u) {
        const int j = P->obs_image[G.kept[G.pt_obs[u]]];
        if (img_col[j] >= 0) f = std::min(f, img_col[j]);
      }
    }
    img_first[i] = f;
  }
  for (int i = 0; i < G.NI; ++i)
    for (int e = 0; e < 6; ++e) if (G.col_pose[i * 6 + e] >= 0) X.first[G.col_pose[i * 6 + e]] = img_first[i];
  // (intrinsics rows: dense, first = 0)
}

// pose part of one kept observation: free pose columns and the 2 x m values
inline int pose_row(const Program& G, const Lin& L, size_t k, int img, int cols[6], double vals[2][6]) {
  int m = 0;
  for (int e = 0; e < 6; ++e) {
    const int col = G.col_pose[img * 6 + e];
    if (col >= 0) { cols[m] = col; vals[0][m] = L.Jc[k * 12 + e]; vals[1][m] = L.Jc[k * 12 + 6 + e]; ++m; }
  }
  return m;
}

// Lower triangle of S (rows >= columns in the reduced order) and v; the upper triangle is left zero.
void schur_eliminate_sparse(const Program& G, const Lin& L, const SparseIndex& X, const std::vector<double>& D,
                            std::vector<double>& S, std::vector<double>& v, std::vector<double>& ete_inv,
                            std::vector<double>& gp) {
  const mavba_problem* P = G.P;
  const int n = G.n_cam;
  S.assign((size_t)n * n, 0.0);
  v.assign(n, 0.0);
  ete_inv.assign((size_t)G.n_fp * 9, 0.0);
  gp.assign((size_t)G.n_fp * 3, 0.0);
  for (int j = 0; j < n; ++j) S[(size_t)j * n + j] = D[j] * D[j];
  for (size_t q = 0; q < G.kept_prior.size(); ++q) {
    const int i = P->rot_prior_image[G.kept_prior[q]];
    for (int a = 0; a < 3; ++a) {
      const int ca = G.col_pose[i * 6 + a];
      if (ca < 0) continue;
      v[ca] += L.pJ[q * 3 + a] * L.pr[q];
      for (int b = 0; b < 3; ++b) {
        const int cb = G.col_pose[i * 6 + b];
        if (cb >= 0 && cb <= ca) S[(size_t)ca * n + cb] += L.pJ[q * 3 + a] * L.pJ[q * 3 + b];
      }
    }
  }
  // pass 1, per point: damped 3x3 block, its inverse, g_p, inv * g_p
  std::vector<double> ig((size_t)G.n_fp * 3, 0.0);
#pragma omp parallel for schedule(static)
  for (int p = 0; p < G.NP; ++p) {
    const int fp = G.idx_point[p];
    if (fp < 0) continue;
    double ete[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int64_t t = G.pt_start[p]; t < G.pt_start[p + 1]; ++t) {
      const size_t k = (size_t)G.pt_obs[t];
      for (int row = 0; row < 2; ++row) {
        const double* e = &L.Jp[k * 6 + row * 3];
        const double rr = L.r[k * 2 + row];
        for (int x = 0; x < 3; ++x) { g[x] += e[x] * rr; for (int y = 0; y < 3; ++y) ete[x * 3 + y] += e[x] * e[y]; }
      }
    }
    for (int x = 0; x < 3; ++x) { const double d = D[n + 3 * fp + x]; ete[x * 4] += d * d; }
    double inv[9];
    if (!invert3(ete, inv)) for (int x = 0; x < 9; ++x) inv[x] = std::numeric_limits<double>::quiet_NaN();
    for (int x = 0; x < 9; ++x) ete_inv[(size_t)fp * 9 + x] = inv[x];
    for (int x = 0; x < 3; ++x) {
      gp[(size_t)fp * 3 + x] = g[x];
      ig[(size_t)fp * 3 + x] = inv[x * 3] * g[0] + inv[x * 3 + 1] * g[1] + inv[x * 3 + 2] * g[2];
    }
  }
  // pass 2, pose rows: image i owns its row block
#pragma omp parallel for schedule(dynamic, 1)
  for (int i = 0; i < G.NI; ++i) {
    for (int64_t t = X.img_start[i]; t < X.img_start[i + 1]; ++t) {
      const size_t ka = (size_t)X.img_obs[t];
      int ca[6]; double fa[2][6];
      const int ma = pose_row(G, L, ka, i, ca, fa);
      if (ma == 0) continue;
      // F^T F and F^T r of this observation (diagonal block, lower part)
      for (int x = 0; x < ma; ++x) {
        v[ca[x]] += fa[0][x] * L.r[ka * 2] + fa[1][x] * L.r[ka * 2 + 1];
        for (int y = 0; y <= x; ++y) S[(size_t)ca[x] * n + ca[y]] += fa[0][x] * fa[0][y] + fa[1][x] * fa[1][y];
      }
      const int p = P->obs_point[G.kept[ka]];
      const int fp = G.idx_point[p];
      if (fp < 0) continue;
      const double* inv = &ete_inv[(size_t)fp * 9];
      // Ta = (F_a^T E_a) inv   (ma x 3)
      double Wa[6][3], Ta[6][3];
      for (int x = 0; x < ma; ++x)
        for (int y = 0; y < 3; ++y) Wa[x][y] = fa[0][x] * L.Jp[ka * 6 + y] + fa[1][x] * L.Jp[ka * 6 + 3 + y];
      for (int x = 0; x < ma; ++x)
        for (int y = 0; y < 3; ++y) Ta[x][y] = Wa[x][0] * inv[y] + Wa[x][1] * inv[3 + y] + Wa[x][2] * inv[6 + y];
      for (int x = 0; x < ma; ++x)
        v[ca[x]] -= Wa[x][0] * ig[(size_t)fp * 3] + Wa[x][1] * ig[(size_t)fp * 3 + 1] + Wa[x][2] * ig[(size_t)fp * 3 + 2];
      for (int64_t u = G.pt_start[p]; u < G.pt_start[p + 1]; ++u) {
        const size_t kb = (size_t)G.pt_obs[u];
        const int j = P->obs_image[G.kept[kb]];
        if (j > i) continue;  // lower triangle: the owner of the later image writes the block
        int cb[6]; double fb[2][6];
        const int mb = pose_row(G, L, kb, j, cb, fb);
        for (int y = 0; y < mb; ++y) {
          const double wb[3] = {fb[0][y] * L.Jp[kb * 6] + fb[1][y] * L.Jp[kb * 6 + 3],
                                fb[0][y] * L.Jp[kb * 6 + 1] + fb[1][y] * L.Jp[kb * 6 + 4],
                                fb[0][y] * L.Jp[kb * 6 + 2] + fb[1][y] * L.Jp[kb * 6 + 5]};
          for (int x = 0; x < ma; ++x)
            if (cb[y] <= ca[x]) S[(size_t)ca[x] * n + cb[y]] -= Ta[x][0] * wb[0] + Ta[x][1] * wb[1] + Ta[x][2] * wb[2];
        }
      }
    }
  }
  // pass 3, intrinsics rows: chunks of points accumulate private strips (K rows x n columns + v), added in chunk order
  for (int c = 0; c < G.NC; ++c) {
    const int kc0 = G.col_intr[c];
    if (kc0 < 0) continue;
    const int K = G.K[c];
    const int nchunk = std::max(1, std::min(256, G.NP / 512));
    std::vector<double> strips((size_t)nchunk * K * (n + 1), 0.0);
#pragma omp parallel for schedule(dynamic, 1)
    for (int ch = 0; ch < nchunk; ++ch) {
      double* strip = &strips[(size_t)ch * K * (n + 1)];
      const int p0 = (int)((int64_t)G.NP * ch / nchunk), p1 = (int)((int64_t)G.NP * (ch + 1) / nchunk);
      std::vector<int> cams;
      for (int p = p0; p < p1; ++p) {
        const int fp = G.idx_point[p];
        double Wk[9][3];
        bool any = false;
        for (int x = 0; x < K; ++x) Wk[x][0] = Wk[x][1] = Wk[x][2] = 0.0;
        for (int64_t t = G.pt_start[p]; t < G.pt_start[p + 1]; ++t) {
          const size_t k = (size_t)G.pt_obs[t];
          const int i = P->obs_image[G.kept[k]];
          if (P->image_camera[i] != c) continue;
          any = true;
          const double* k0 = &L.Jk[k * 18];
          const double* k1 = &L.Jk[k * 18 + 9];
          int ca[6]; double fa[2][6];
          const int ma = pose_row(G, L, k, i, ca, fa);
          for (int x = 0; x < K; ++x) {
            double* row = strip + (size_t)x * (n + 1);
            row[n] += k0[x] * L.r[k * 2] + k1[x] * L.r[k * 2 + 1];
            for (int y = 0; y < ma; ++y) row[ca[y]] += k0[x] * fa[0][y] + k1[x] * fa[1][y];
            for (int y = 0; y <= x; ++y) row[kc0 + y] += k0[x] * k0[y] + k1[x] * k1[y];
            for (int y = 0; y < 3; ++y) Wk[x][y] += k0[x] * L.Jp[k * 6 + y] + k1[x] * L.Jp[k * 6 + 3 + y];
          }
        }
        if (!any || fp < 0) continue;
        const double* inv = &ete_inv[(size_t)fp * 9];
        double Tk[9][3];
        for (int x = 0; x < K; ++x)
          for (int y = 0; y < 3; ++y) Tk[x][y] = Wk[x][0] * inv[y] + Wk[x][1] * inv[3 + y] + Wk[x][2] * inv[6 + y];
        for (int x = 0; x < K; ++x)
          strip[(size_t)x * (n + 1) + n] -= Wk[x][0] * ig[(size_t)fp * 3] + Wk[x][1] * ig[(size_t)fp * 3 + 1] + Wk[x][2] * ig[(size_t)fp * 3 + 2];
        // against every pose block of the point, and against the intrinsics blocks of cameras c' <= c it is seen by
        cams.clear();
        for (int64_t u = G.pt_start[p]; u < G.pt_start[p + 1]; ++u) {
          const size_t kb = (size_t)G.pt_obs[u];
          const int j = P->obs_image[G.kept[kb]];
          int cb[6]; double fb[2][6];
          const int mb = pose_row(G, L, kb, j, cb, fb);
          for (int y = 0; y < mb; ++y) {
            const double wb[3] = {fb[0][y] * L.Jp[kb * 6] + fb[1][y] * L.Jp[kb * 6 + 3],
                                  fb[0][y] * L.Jp[kb * 6 + 1] + fb[1][y] * L.Jp[kb * 6 + 4],
                                  fb[0][y] * L.Jp[kb * 6 + 2] + fb[1][y] * L.Jp[kb * 6 + 5]};
            for (int x = 0; x < K; ++x) strip[(size_t)x * (n + 1) + cb[y]] -= Tk[x][0] * wb[0] + Tk[x][1] * wb[1] + Tk[x][2] * wb[2];
          }
          const int c2 = P->image_camera[j];
          if (c2 <= c && G.col_intr[c2] >= 0 && std::find(cams.begin(), cams.end(), c2) == cams.end()) cams.push_back(c2);
        }
        for (int c2 : cams) {
          double W2[9][3];
          const int K2 = G.K[c2];
          for (int y = 0; y < K2; ++y) W2[y][0] = W2[y][1] = W2[y][2] = 0.0;
          for (int64_t u = G.pt_start[p]; u < G.pt_start[p + 1]; ++u) {
            const size_t kb = (size_t)G.pt_obs[u];
            if (P->image_camera[P->obs_image[G.kept[kb]]] != c2) continue;
            for (int y = 0; y < K2; ++y)
              for (int z = 0; z < 3; ++z) W2[y][z] += L.Jk[kb * 18 + y] * L.Jp[kb * 6 + z] + L.Jk[kb * 18 + 9 + y] * L.Jp[kb * 6 + 3 + z];
          }
          for (int x = 0; x < K; ++x)
            for (int y = 0; y < K2; ++y) {
              if (c2 == c && y > x) continue;
              strip[(size_t)x * (n + 1) + G.col_intr[c2] + y] -= Tk[x][0] * W2[y][0] + Tk[x][1] * W2[y][1] + Tk[x][2] * W2[y][2];
            }
        }
      }
    }
#pragma omp parallel for schedule(static)
    for (int col = 0; col <= n; ++col)
      for (int x = 0; x < K; ++x) {
        double acc = 0.0;
        for (int ch = 0; ch < nchunk; ++ch) acc += strips[((size_t)ch * K + x) * (n + 1) + col];
        if (col == n) v[kc0 + x] += acc;
        else if (col <= kc0 + x) S[(size_t)(kc0 + x) * n + col] += acc;
      }
  }
}

// Blocked Cholesky (lower, in place, row-major) that never leaves the row envelope: row i is structurally zero
// left of first[i], and so is its row of the factor. Returns false if not SPD.
bool envelope_cholesky(int n, double* A, const std::vector<int>& first) {
  const int NB = 64;
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int kb = std::min(NB, n - k0);
    for (int j = k0; j < k0 + kb; ++j) {
      const int lo = std::max(k0, first[j]);
      double d = A[(size_t)j * n + j];
      for (int k = lo; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
      if (!(d > 0.0) || !std::isfinite(d)) return false;
      d = std::sqrt(d);
      A[(size_t)j * n + j] = d;
      for (int i = j + 1; i < k0 + kb; ++i) {
        if (first[i] > j) continue;
        double s = A[(size_t)i * n + j];
        for (int k = std::max(lo, first[i]); k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
        A[(size_t)i * n + j] = s / d;
      }
    }
    const int r0 = k0 + kb;
    // rows that reach into this panel
    std::vector<int> act;
    for (int i = r0; i < n; ++i) if (first[i] < r0) act.push_back(i);
    const int na = (int)act.size();
#pragma omp parallel for schedule(static)
    for (int t = 0; t < na; ++t) {
      const int i = act[t];
      for (int j = std::max(k0, first[i]); j < k0 + kb; ++j) {
        double s = A[(size_t)i * n + j];
        for (int k = std::max(std::max(k0, first[i]), first[j]); k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
        A[(size_t)i * n + j] = s / A[(size_t)j * n + j];
      }
    }
#pragma omp parallel for schedule(dynamic, 8)
    for (int t = 0; t < na; ++t) {
      const int i = act[t];
      const int ki = std::max(k0, first[i]);
      const double* ai = &A[(size_t)i * n];
      for (int u = 0; u <= t; ++u) {
        const int j = act[u];
        const double* aj = &A[(size_t)j * n];
        double s = 0.0;
        for (int k = std::max(ki, first[j]); k < k0 + kb; ++k) s += ai[k] * aj[k];
        A[(size_t)i * n + j] -= s;
      }
    }
  }
  return true;
}

void envelope_solve(int n, const double* L, const std::vector<int>& first, double* b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = first[i]; k < i; ++k) s -= L[(size_t)i * n + k] * b[k];
    b[i] = s / L[(size_t)i * n + i];
  }
  // backward, column-oriented so that only the envelope is read: x_i known -> subtract its column from the rows above
  for (int i = n - 1; i >= 0; --i) {
    b[i] /= L[(size_t)i * n + i];
    const double xi = b[i];
    for (int k = first[i]; k < i; ++k) b[k] -= L[(size_t)i * n + k] * xi;
  }
}


// out[col] += sum over the rows of J of J[row][col] * (squares ? J[row][col] : r[row]), every column owned by one
// thread (pose columns: the image; point columns: the point) or summed from fixed chunks in chunk order (intrinsics):
// the result does not depend on the thread count. Mode-1 counterpart of the serial loops in evaluate / column_sq_norms.
void accumulate_columns_parallel(const Program& G, const Lin& L, bool squares, std::vector<double>& out) {
  const mavba_problem* P = G.P;
  SparseIndex X(G);
  build_sparse_index(G, X);
  out.assign((size_t)G.n_cam + 3 * (size_t)G.n_fp, 0.0);
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < G.NI; ++i)
    for (int64_t t = X.img_start[i]; t < X.img_start[i + 1]; ++t) {
      const size_t k = (size_t)X.img_obs[t];
      for (int row = 0; row < 2; ++row)
        for (int e = 0; e < 6; ++e) {
          const int col = G.col_pose[i * 6 + e];
          if (col < 0) continue;
          const double x = L.Jc[k * 12 + row * 6 + e];
          out[col] += x * (squares ? x : L.r[k * 2 + row]);
        }
    }
#pragma omp parallel for schedule(static)
  for (int p = 0; p < G.NP; ++p) {
    const int fp = G.idx_point[p];
    if (fp < 0) continue;
    for (int64_t t = G.pt_start[p]; t < G.pt_start[p + 1]; ++t) {
      const size_t k = (size_t)G.pt_obs[t];
      for (int row = 0; row < 2; ++row)
        for (int e = 0; e < 3; ++e) {
          const double x = L.Jp[k * 6 + row * 3 + e];
          out[G.n_cam + 3 * fp + e] += x * (squares ? x : L.r[k * 2 + row]);
        }
    }
  }
  const int nchunk = 256;
  const int64_t nk = (int64_t)G.kept.size();
  std::vector<double> part((size_t)nchunk * G.NC * 9, 0.0);
#pragma omp parallel for schedule(dynamic, 1)
  for (int ch = 0; ch < nchunk; ++ch)
    for (int64_t k = nk * ch / nchunk; k < nk * (ch + 1) / nchunk; ++k) {
      const int c = P->image_camera[P->obs_image[G.kept[k]]];
      if (G.col_intr[c] < 0) continue;
      double* acc = &part[((size_t)ch * G.NC + c) * 9];
      for (int row = 0; row < 2; ++row)
        for (int e = 0; e < G.K[c]; ++e) {
          const double x = L.Jk[k * 18 + row * 9 + e];
          acc[e] += x * (squares ? x : L.r[k * 2 + row]);
        }
    }
  for (int c = 0; c < G.NC; ++c)
    if (G.col_intr[c] >= 0)
      for (int e = 0; e < G.K[c]; ++e)
        for (int ch = 0; ch < nchunk; ++ch) out[G.col_intr[c] + e] += part[((size_t)ch * G.NC + c) * 9 + e];
  for (size_t q = 0; q < G.kept_prior.size(); ++q) {
    const int i = P->rot_prior_image[G.kept_prior[q]];
    for (int e = 0; e < 3; ++e) {
      const int col = G.col_pose[i * 6 + e];
      if (col >= 0) out[col] += L.pJ[q * 3 + e] * (squares ? L.pJ[q * 3 + e] : L.pr[q]);
    }
  }
}

// One LM linear solve (LevenbergMarquardtStrategy::ComputeStep + SchurComplementSolver).
// Lsc is the *scaled* linearisation, diag the clamped squared column norms.
// Returns false on linear-solver failure. step = -y over [cameras|points].
bool compute_step(const Program& G, const Lin& Lsc, const std::vector<double>& diag, double radius,
                  std::vector<double>& step, std::vector<double>* S_out, std::vector<double>* v_out) {
  const int n = G.n_cam;
  const size_t np = (size_t)n + 3 * (size_t)G.n_fp;
  std::vector<double> D(np);
  for (size_t j = 0; j < np; ++j) D[j] = std::sqrt(diag[j] / radius);
  std::vector<double> S, v, ete_inv, gp;
  step.assign(np, 0.0);
  std::vector<double> y;
  if (g_linear_solver_mode == 1) {
    SparseIndex X(G);
    build_sparse_index(G, X);
    schur_eliminate_sparse(G, Lsc, X, D, S, v, ete_inv, gp);
    if (S_out) {
      *S_out = S;
      for (int a = 0; a < n; ++a) for (int b = a + 1; b < n; ++b) (*S_out)[(size_t)a * n + b] = S[(size_t)b * n + a];
    }
    if (v_out) *v_out = v;
    y = v;
    if (n > 0) {
      if (!envelope_cholesky(n, S.data(), X.first)) return false;
      envelope_solve(n, S.data(), X.first, y.data());
    }
  } else {
    schur_eliminate(G, Lsc, D, S, v, ete_inv, gp);
    if (S_out) *S_out = S;
    if (v_out) *v_out = v;
    y = v;
    if (n > 0) {
      if (!dense_cholesky(n, S.data())) return false;
      cholesky_solve(n, S.data(), y.data());
    }
  }
  std::vector<double> yp(3 * (size_t)G.n_fp, 0.0);
  back_substitute(G, Lsc, ete_inv, gp, y.data(), yp.data());
  for (int j = 0; j < n; ++j) step[j] = -y[j];
  for (size_t j = 0; j < yp.size(); ++j) step[n + j] = -yp[j];
  for (size_t j = 0; j < np; ++j) if (!std::isfinite(step[j])) return false;
  return true;
}

// model_cost_change = -m.(r + m/2), m = J step   (trust_region_minimizer.cc)
double model_cost_change(const Program& G, const Lin& Lsc, const std::vector<double>& step) {
  const mavba_problem* P = G.P;
  const int n = G.n_cam;
  double acc = 0.0;
  auto one = [&](size_t k, double& a) {
    const int p = P->obs_point[G.kept[k]];
    const int fp = G.idx_point[p];
    for (int row = 0; row < 2; ++row) {
      int cols[15]; double vals[15];
      const int m = camera_row(G, Lsc, k, row, cols, vals);
      double mr = 0.0;
      for (int x = 0; x < m; ++x) mr += vals[x] * step[cols[x]];
      if (fp >= 0) for (int x = 0; x < 3; ++x) mr += Lsc.Jp[k * 6 + row * 3 + x] * step[n + 3 * fp + x];
      a += mr * (Lsc.r[k * 2 + row] + mr / 2.0);
    }
  };
  if (g_linear_solver_mode == 1) {
    const int nchunk = 256;
    const int64_t nk = (int64_t)G.kept.size();
    double part[256];
#pragma omp parallel for schedule(dynamic, 1)
    for (int ch = 0; ch < nchunk; ++ch) {
      double a = 0.0;
      for (int64_t k = nk * ch / nchunk; k < nk * (ch + 1) / nchunk; ++k) one((size_t)k, a);
      part[ch] = a;
    }
    for (int ch = 0; ch < nchunk; ++ch) acc += part[ch];
  } else {
    for (size_t k = 0; k < G.kept.size(); ++k) one(k, acc);
  }
  for (size_t q = 0; q < G.kept_prior.size(); ++q) {
    const int i = P->rot_prior_image[G.kept_prior[q]];
    double mr = 0.0;
    for (int e = 0; e < 3; ++e) { const int col = G.col_pose[i * 6 + e]; if (col >= 0) mr += Lsc.pJ[q * 3 + e] * step[col]; }
    acc += mr * (Lsc.pr[q] + mr / 2.0);
  }
  return -acc;
}

double max_abs(const std::vector<double>& x) {
  double m = 0.0;
  for (double v : x) m = std::max(m, std::fabs(v));
  return m;
}

void clamp_diag(const mavba_options& o, std::vector<double>& d) {
  for (double& x : d) x = std::min(std::max(x, o.min_lm_diagonal), o.max_lm_diagonal);
}

}  // namespace

// ===========================================================================
// extern "C" surface (loaded through ctypes by tests/ and bench.py only)
// ===========================================================================
extern "C" {

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}

// 0 = dense Schur complement + dense Cholesky (default), 1 = block-sparse Schur complement + envelope Cholesky
void oracle_set_linear_solver(int mode) { g_linear_solver_mode = mode == 1 ? 1 : 0; }
int oracle_get_linear_solver(void) { return g_linear_solver_mode; }

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_options_init(mavba_options* o) {
  std::memset(o, 0, sizeof(*o));
  // BundleAdjustmentOptions defaults, bundle_adjustment.h:40-50
  o->max_num_iterations = 100;
  o->function_tolerance = 1e-4;
  o->gradient_tolerance = 1e-8;
  o->loss_scale_factor = 1.0;
  o->update_point_errors = 0;
  o->print_progress = 0;
  // Ceres 1.8 Solver::Options defaults (never overridden by the reference)
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 10;  // bundle_adjustment.cc:559
  o->jacobi_scaling = 1;
  o->device = -1;
  o->profile_kernels = 0;
}

void oracle_world2image(int model, const double* params, double x, double y, double z,
                        double* u, double* v) {
  world2image<double>(model, x, y, z, *u, *v, params);
}

void oracle_image2world(int model, const double* params, double u, double v, double* x,
                        double* y, double* z) {
  image2world(model, u, v, *x, *y, *z, params);
}

// Rotate a point: ceres::AngleAxisRotatePoint.
void oracle_rotate_point(const double* rvec, const double* pt, double* out) {
  angle_axis_rotate_point<double>(rvec, pt, out);
}

void oracle_rotation_matrix(const double* rvec, double* R_colmajor) {
  angle_axis_to_rotation_matrix<double>(rvec, R_colmajor);
}

// Raw (no loss) residual + Jacobian of one observation. mode 0 = Jets, 1 = analytic.
void oracle_obs_jacobian(int mode, int model, const double* pose, const double* X,
                         const double* cam, const double* uv, double* r, double* Jc,
                         double* Jp, double* Jk) {
  obs_jacobian(mode, model, pose, X, cam, uv, r, Jc, Jp, Jk);
}

// Rotation-prior residual and its 1x3 Jacobian (Jets).
void oracle_rot_prior(const double* rvec, const double* rvec0, double weight, double* res,
                      double* jac3) {
  double R0[9];
  angle_axis_to_rotation_matrix<double>(rvec0, R0);
  typedef Jet<3> J;
  J rv[3] = {J(rvec[0], 0), J(rvec[1], 1), J(rvec[2], 2)};
  const J r = rotation_prior_residual<J>(rv, R0, weight);
  *res = r.a;
  for (int e = 0; e < 3; ++e) jac3[e] = r.v[e];
}

// Loss-corrected residuals + Jacobians for every observation, caller order
// (dropped all-constant residual blocks are reported as zeros). Returns cost.
int oracle_eval_jacobian(const mavba_problem* P, const mavba_options* opt, int jac_mode,
                         double* cost, double* r, double* Jc, double* Jp, double* Jk) {
  Program G;
  const int rc = build_program(G, P, opt, jac_mode);
  if (rc != MAVBA_OK) return rc;
  Lin L;
  double c = 0.0;
  evaluate(G, G.poses.data(), G.intr.data(), G.points.data(), true, L, &c);
  if (cost) *cost = c + G.fixed_cost;
  if (r) std::memset(r, 0, sizeof(double) * 2 * (size_t)G.NO);
  if (Jc) std::memset(Jc, 0, sizeof(double) * 12 * (size_t)G.NO);
  if (Jp) std::memset(Jp, 0, sizeof(double) * 6 * (size_t)G.NO);
  if (Jk) std::memset(Jk, 0, sizeof(double) * 18 * (size_t)G.NO);
  for (size_t k = 0; k < G.kept.size(); ++k) {
    const size_t o = (size_t)G.kept[k];
    if (r) std::memcpy(&r[o * 2], &L.r[k * 2], sizeof(double) * 2);
    if (Jc) std::memcpy(&Jc[o * 12], &L.Jc[k * 12], sizeof(double) * 12);
    if (Jp) std::memcpy(&Jp[o * 6], &L.Jp[k * 6], sizeof(double) * 6);
    if (Jk) std::memcpy(&Jk[o * 18], &L.Jk[k * 18], sizeof(double) * 18);
  }
  return MAVBA_OK;
}

int oracle_reduced_dim(const mavba_problem* P) { return 6 * P->num_images + 9 * P->num_cameras; }

// First-iteration LM linear system at `radius`, expanded to the device's
// uniform indexing (pose column 6*i+e, intrinsic 6*NI + 9*c + e; constant or
// absent columns: S_jj = 1, v_j = 0), and the un-scaled step it yields.
int oracle_linear_step(const mavba_problem* P, const mavba_options* opt, int jac_mode,
                       double radius, double* S_full, double* v_full, double* d_poses,
                       double* d_intr, double* d_points, double* model_change) {
  Program G;
  const int rc = build_program(G, P, opt, jac_mode);
  if (rc != MAVBA_OK) return rc;
  Lin L;
  evaluate(G, G.poses.data(), G.intr.data(), G.points.data(), true, L, nullptr);
  const size_t np = (size_t)G.n_cam + 3 * (size_t)G.n_fp;
  std::vector<double> scale(np, 1.0), diag;
  if (opt->jacobi_scaling) {
    column_sq_norms(G, L, scale);
    for (double& s : scale) s = 1.0 / (1.0 + std::sqrt(s));
    scale_columns(G, L, scale);
  }
  column_sq_norms(G, L, diag);
  clamp_diag(*opt, diag);
  std::vector<double> step, S, v;
  const bool ok = compute_step(G, L, diag, radius, step, &S, &v);
  const int nf = oracle_reduced_dim(P);
  std::vector<int> map(G.n_cam, -1);
  for (int i = 0; i < G.NI; ++i)
    for (int e = 0; e < 6; ++e) if (G.col_pose[i * 6 + e] >= 0) map[G.col_pose[i * 6 + e]] = 6 * i + e;
  for (int c = 0; c < G.NC; ++c)
    if (G.col_intr[c] >= 0) for (int e = 0; e < G.K[c]; ++e) map[G.col_intr[c] + e] = 6 * G.NI + 9 * c + e;
  if (S_full) {
    std::memset(S_full, 0, sizeof(double) * (size_t)nf * nf);
    for (int j = 0; j < nf; ++j) S_full[(size_t)j * nf + j] = 1.0;
    for (int a = 0; a < G.n_cam; ++a)
      for (int b = 0; b < G.n_cam; ++b) S_full[(size_t)map[a] * nf + map[b]] = S[(size_t)a * G.n_cam + b];
  }
  if (v_full) {
    std::memset(v_full, 0, sizeof(double) * nf);
    for (int a = 0; a < G.n_cam; ++a) v_full[map[a]] = v[a];
  }
  if (!ok) return MAVBA_ERR_INVALID_ARGUMENT;
  if (model_change) *model_change = model_cost_change(G, L, step);
  std::vector<double> delta(np);
  for (size_t j = 0; j < np; ++j) delta[j] = step[j] * scale[j];
  if (d_poses) {
    std::memset(d_poses, 0, sizeof(double) * 6 * (size_t)G.NI);
    for (int i = 0; i < G.NI * 6; ++i) if (G.col_pose[i] >= 0) d_poses[i] = delta[G.col_pose[i]];
  }
  if (d_intr) {
    std::memset(d_intr, 0, sizeof(double) * 9 * (size_t)G.NC);
    for (int c = 0; c < G.NC; ++c)
      if (G.col_intr[c] >= 0) for (int e = 0; e < G.K[c]; ++e) d_intr[c * 9 + e] = delta[G.col_intr[c] + e];
  }
  if (d_points) {
    std::memset(d_points, 0, sizeof(double) * 3 * (size_t)G.NP);
    for (int p = 0; p < G.NP; ++p)
      if (G.idx_point[p] >= 0) for (int e = 0; e < 3; ++e) d_points[p * 3 + e] = delta[G.n_cam + 3 * G.idx_point[p] + e];
  }
  return MAVBA_OK;
}

/*
 * The full solve: ceres::Solve with TRUST_REGION / LEVENBERG_MARQUARDT /
 * SPARSE_SCHUR as configured at bundle_adjustment.cc:553-569, restated from
 * Ceres 1.8 trust_region_minimizer.cc. Writes parameters back in place unless
 * the solve ends in NUMERICAL_FAILURE (Ceres leaves user state untouched then).
 * `max_iters_override` >= 0 caps the iteration count (cpu_baseline sampling).
 * `iter_seconds` (may be NULL) receives the wall time spent inside the loop.
 */
int oracle_solve_ex(const mavba_problem* P, const mavba_options* opt, int jac_mode,
                    mavba_result* res, double* point_error, double* iter_seconds) {
  const auto t_begin = std::chrono::steady_clock::now();
  Program G;
  const int rc = build_program(G, P, opt, jac_mode);
  if (rc != MAVBA_OK) return rc;
  std::memset(res, 0, sizeof(*res));
  res->num_residuals = G.num_residuals;
  res->num_residuals_reduced = G.num_residuals_reduced;
  res->num_parameters_reduced = G.num_parameters_reduced;
  res->fixed_cost = G.fixed_cost;
  res->termination = MAVBA_TERM_NO_CONVERGENCE;

  const size_t np = (size_t)G.n_cam + 3 * (size_t)G.n_fp;
  std::vector<double>&x_poses = G.poses, &x_intr = G.intr, &x_points = G.points;
  Lin L;
  double cost = 0.0;
  bool failed = false;
  const auto t_loop = std::chrono::steady_clock::now();

  if (G.num_residuals_reduced == 0 || np == 0) {
    // nothing to optimise: ceres returns with cost = fixed cost
    evaluate(G, x_poses.data(), x_intr.data(), x_points.data(), false, L, &cost);
    res->initial_cost = res->final_cost = cost + G.fixed_cost;
    res->termination = MAVBA_TERM_FUNCTION_TOLERANCE;
  } else {
    double x_norm = reduced_norm(G, x_poses, x_intr, x_points);
    evaluate(G, x_poses.data(), x_intr.data(), x_points.data(), true, L, &cost);
    res->initial_cost = cost + G.fixed_cost;
    double grad_max = max_abs(L.grad);
    const double init_grad_max = std::max(grad_max, std::numeric_limits<double>::epsilon());
    const double abs_gtol = opt->gradient_tolerance * init_grad_max;
    std::vector<double> scale(np, 1.0), diag, step, delta(np);
    std::vector<double> c_poses, c_intr, c_points;
    double radius = opt->initial_trust_region_radius;
    double decrease_factor = 2.0;
    bool reuse_diagonal = false;
    int invalid = 0, iteration = 0;
    bool done = false;
    if (grad_max <= abs_gtol) { res->termination = MAVBA_TERM_GRADIENT_TOLERANCE; done = true; }
    if (!done) {
      if (opt->jacobi_scaling) {
        column_sq_norms(G, L, scale);
        for (double& s : scale) s = 1.0 / (1.0 + std::sqrt(s));
        scale_columns(G, L, scale);
      }
    }
    if (opt->print_progress && !done)
      std::printf("%4s %14s %12s %10s %10s %10s %10s\n", "iter", "cost", "cost_change", "|grad|", "|step|", "tr_ratio", "tr_radius");
    if (opt->print_progress && !done)
      std::printf("%4d %14.6e %12.2e %10.2e %10.2e %10.2e %10.2e\n", 0, cost + G.fixed_cost, 0.0, grad_max, 0.0, 0.0, radius);
    while (!done) {
      if (iteration >= opt->max_num_iterations) { res->termination = MAVBA_TERM_NO_CONVERGENCE; break; }
      ++iteration;
      if (!reuse_diagonal) { column_sq_norms(G, L, diag); clamp_diag(*opt, diag); }
      const bool solved = compute_step(G, L, diag, radius, step, nullptr, nullptr);
      reuse_diagonal = true;
      bool valid = false, successful = false;
      double mcc = 0.0, new_cost = 0.0, rel = 0.0, step_norm = 0.0;
      if (solved) {
        mcc = model_cost_change(G, L, step);
        valid = !(mcc < 0.0);
      }
      if (!valid) {
        if (++invalid >= opt->max_num_consecutive_invalid_steps) {
          res->termination = MAVBA_TERM_NUMERICAL_FAILURE; failed = true; break;
        }
      } else {
        invalid = 0;
        for (size_t j = 0; j < np; ++j) delta[j] = step[j] * scale[j];
        plus(G, x_poses, x_intr, x_points, delta.data(), c_poses, c_intr, c_points);
        evaluate(G, c_poses.data(), c_intr.data(), c_points.data(), false, L, &new_cost);
        // step_norm = |x - x_plus_delta| over the reduced vector
        double sn = 0.0;
        for (size_t j = 0; j < np; ++j) sn += delta[j] * delta[j];
        step_norm = std::sqrt(sn);
        if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
          res->termination = MAVBA_TERM_PARAMETER_TOLERANCE; break;
        }
        const double cost_change = cost - new_cost;
        if (std::fabs(cost_change) < opt->function_tolerance * cost) {
          res->termination = MAVBA_TERM_FUNCTION_TOLERANCE; break;
        }
        rel = cost_change / mcc;
        successful = rel > opt->min_relative_decrease;
      }
      if (successful) {
        ++res->num_successful_steps;
        // LevenbergMarquardtStrategy::StepAccepted
        radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
        radius = std::min(opt->max_trust_region_radius, radius);
        decrease_factor = 2.0;
        reuse_diagonal = false;
        x_poses.swap(c_poses); x_intr.swap(c_intr); x_points.swap(c_points);
        x_norm = reduced_norm(G, x_poses, x_intr, x_points);
        evaluate(G, x_poses.data(), x_intr.data(), x_points.data(), true, L, &cost);
        grad_max = max_abs(L.grad);
        if (grad_max <= abs_gtol) {
          res->termination = MAVBA_TERM_GRADIENT_TOLERANCE;
          if (opt->print_progress)
            std::printf("%4d %14.6e %12.2e %10.2e %10.2e %10.2e %10.2e\n", iteration, cost + G.fixed_cost, 0.0, grad_max, step_norm, rel, radius);
          break;
        }
        if (opt->jacobi_scaling) scale_columns(G, L, scale);
      } else {
        ++res->num_unsuccessful_steps;
        // StepRejected / StepIsInvalid
        radius = radius / decrease_factor;
        decrease_factor *= 2.0;
        reuse_diagonal = true;
      }
      if (opt->print_progress)
        std::printf("%4d %14.6e %12.2e %10.2e %10.2e %10.2e %10.2e\n", iteration, cost + G.fixed_cost,
                    valid ? (successful ? 0.0 : 0.0) : 0.0, grad_max, step_norm, rel, radius);
      if (radius < opt->min_trust_region_radius) { res->termination = MAVBA_TERM_PARAMETER_TOLERANCE; break; }
    }
    res->final_cost = cost + G.fixed_cost;
    res->final_gradient_max_norm = grad_max;
    res->final_trust_region_radius = radius;
  }
  const auto t_end = std::chrono::steady_clock::now();
  res->setup_seconds = std::chrono::duration<double>(t_loop - t_begin).count();
  res->solve_seconds = std::chrono::duration<double>(t_end - t_loop).count();
  if (iter_seconds) *iter_seconds = res->solve_seconds;

  if (!failed) {
    std::memcpy(P->poses, x_poses.data(), sizeof(double) * x_poses.size());
    std::memcpy(P->intrinsics, x_intr.data(), sizeof(double) * x_intr.size());
    std::memcpy(P->points, x_points.data(), sizeof(double) * x_points.size());
  }

  // point3D_errors — bundle_adjustment.cc:575-598: trivial loss, every
  // residual block in caller order, divided by the point's observation count
  // inside the problem.
  if (point_error && opt->update_point_errors) {
    std::vector<int64_t> cnt(G.NP, 0);
    for (int64_t o = 0; o < G.NO; ++o) cnt[P->obs_point[o]]++;
    for (int64_t o = 0; o < G.NO; ++o) point_error[P->obs_point[o]] = 0.0;
    for (int64_t o = 0; o < G.NO; ++o) {
      const int i = P->obs_image[o], p = P->obs_point[o], c = P->image_camera[i];
      double r[2];
      obs_residual(P->camera_model[c], &P->poses[i * 6], &P->points[p * 3], &P->intrinsics[c * 9],
                   &P->obs_uv[o * 2], r);
      point_error[p] += std::sqrt(r[0] * r[0] + r[1] * r[1]) / (double)cnt[p];
    }
  }
  return MAVBA_OK;
}

int oracle_solve(const mavba_problem* P, const mavba_options* opt, mavba_result* res,
                 double* point_error) {
  return oracle_solve_ex(P, opt, 0, res, point_error, nullptr);
}

// Dense SPD solve (for testing the device Cholesky against the same maths).
int oracle_dense_spd_solve(int n, const double* A, const double* b, double* x) {
  std::vector<double> M(A, A + (size_t)n * n);
  if (!dense_cholesky(n, M.data())) return MAVBA_ERR_INVALID_ARGUMENT;
  std::memcpy(x, b, sizeof(double) * n);
  cholesky_solve(n, M.data(), x);
  return MAVBA_OK;
}

}  // extern "C"
