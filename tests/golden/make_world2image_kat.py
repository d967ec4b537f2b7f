"""Generator of tests/golden/world2image_kat.json and tests/golden/world2image_jet_kat.json.

Runs ONLY in the build container (it reads /root/reference; nothing of the reference travels): a small driver is
compiled against the reference's OWN header src/base3d/camera_models.h and its templates
{Pinhole,OpenCV,Cata}CameraModel::world2image<T> are instantiated with T = double (known answers of the projection,
the eight vectors SURVEY.md 8(c) recorded at the parameter sets of camera_models_test.cc:60-82 plus 100 seeded
(Xc, kappa) per model) and with T = a forward-mode dual number defined in the driver (golden d(u,v)/d(Xc, kappa)
rows for the same samples: what ceres::AutoDiffCostFunction differentiates in the reference).

The header includes <Eigen/Core> for ONE unrelated signature (camera_model_image2world takes std::vector<Eigen::Vector2d>);
the templates under test use no Eigen. This image has no Eigen, so the driver is compiled with -I tests/stubs, whose
Eigen/Core test double only supplies that type name. This is a FIXTURE GENERATOR, not an oracle/_ref build: its output
is data (inputs and expected outputs) committed under tests/golden/ and checked by tests/test_oracle.py against
oracle/ba_oracle.cpp and by tests/test_host_math.py against the kernel maths (csrc/ba_math.h)."""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"

DRIVER = r"""
#include <cmath>
#include <cstdio>
#include <random>
#include "base3d/camera_models.h"

// forward-mode dual number over N directions (the role ceres::Jet plays in the reference)
template <int N> struct Dual {
  double a; double v[N];
  Dual() : a(0) { for (int i = 0; i < N; ++i) v[i] = 0; }
  Dual(double x) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0; }
};
#define BIN(op, EXPR_A, EXPR_V) \
  template <int N> Dual<N> operator op(const Dual<N>& x, const Dual<N>& y) { Dual<N> r; r.a = EXPR_A; for (int i = 0; i < N; ++i) r.v[i] = EXPR_V; return r; }
BIN(+, x.a + y.a, x.v[i] + y.v[i])
BIN(-, x.a - y.a, x.v[i] - y.v[i])
BIN(*, x.a * y.a, x.v[i] * y.a + x.a * y.v[i])
BIN(/, x.a / y.a, (x.v[i] - (x.a / y.a) * y.v[i]) / y.a)
template <int N> Dual<N>& operator+=(Dual<N>& x, const Dual<N>& y) { x = x + y; return x; }
template <int N> Dual<N> sqrt(const Dual<N>& x) { Dual<N> r; r.a = std::sqrt(x.a); for (int i = 0; i < N; ++i) r.v[i] = x.v[i] / (2.0 * r.a); return r; }

template <class Model, int K> void emit(const char* name, int code, const double* X, const double* k, bool last) {
  double u, v;
  Model::template world2image<double>(X[0], X[1], X[2], u, v, k);
  typedef Dual<3 + K> D;
  D x(X[0]), y(X[1]), z(X[2]), du, dv, kk[K];
  x.v[0] = 1; y.v[1] = 1; z.v[2] = 1;
  for (int i = 0; i < K; ++i) { kk[i] = D(k[i]); kk[i].v[3 + i] = 1; }
  Model::template world2image<D>(x, y, z, du, dv, kk);
  std::printf("  {\"model\": \"%s\", \"code\": %d, \"Xc\": [%.17g, %.17g, %.17g], \"params\": [", name, code, X[0], X[1], X[2]);
  for (int i = 0; i < K; ++i) std::printf("%s%.17g", i ? ", " : "", k[i]);
  std::printf("], \"uv\": [%.17g, %.17g], \"du\": [", u, v);
  for (int i = 0; i < 3 + K; ++i) std::printf("%s%.17g", i ? ", " : "", du.v[i]);
  std::printf("], \"dv\": [");
  for (int i = 0; i < 3 + K; ++i) std::printf("%s%.17g", i ? ", " : "", dv.v[i]);
  std::printf("]}%s\n", last ? "" : ",");
}

int main() {
  // the parameter sets of camera_models_test.cc:60-82 and the two points of SURVEY.md 8(c)
  const double P[9] = {651.123, 655.123, 386.123, 511.123, -0.471, 0.223, -0.001, 0.001, 0.0};
  const double pts[2][3] = {{0.5, 0.23, 1.0}, {0.3, -0.2, 2.5}};
  std::printf("[\n");
  for (int q = 0; q < 2; ++q) {
    double k[9];
    for (int i = 0; i < 9; ++i) k[i] = P[i];
    emit<PinholeCameraModel, 4>("PINHOLE", 1, pts[q], k, false);
    emit<OpenCVCameraModel, 8>("OPENCV", 2, pts[q], k, false);
    const double xis[3] = {0.0, 0.5, 1.0};
    for (int t = 0; t < 3; ++t) { k[8] = xis[t]; emit<CataCameraModel, 9>("CATA", 3, pts[q], k, false); }
  }
  // 100 seeded samples per model: points in front of the camera, intrinsics around the synthetic configs'
  std::mt19937_64 rng(20260927);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (int s = 0; s < 100; ++s) {
    const double z = 2.0 + 30.0 * (0.5 + 0.5 * U(rng));
    const double X[3] = {0.45 * z * U(rng), 0.3 * z * U(rng), z};
    double k[9] = {600.0 * (1 + 0.1 * U(rng)), 600.0 * (1 + 0.1 * U(rng)), 376.0 + 20 * U(rng), 240.0 + 20 * U(rng),
                   -0.1 + 0.1 * U(rng), 0.02 + 0.02 * U(rng), 1e-3 * U(rng), 1e-3 * U(rng), 0.3 + 0.3 * U(rng)};
    emit<PinholeCameraModel, 4>("PINHOLE", 1, X, k, false);
    emit<OpenCVCameraModel, 8>("OPENCV", 2, X, k, false);
    emit<CataCameraModel, 9>("CATA", 3, X, k, s == 99);
  }
  std::printf("]\n");
  return 0;
}
"""


def main():
    if not os.path.isdir(REF):
        sys.exit("make_world2image_kat.py runs in the build container only (needs /root/reference)")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "kat.cpp"), os.path.join(d, "kat")
        open(src, "w").write(DRIVER)
        # -O0 and no FMA contraction: the recorded answers are the plain IEEE evaluation of the reference's expressions
        subprocess.check_call(["g++", "-std=c++11", "-O0", "-ffp-contract=off", "-I", os.path.join(ROOT, "tests", "stubs"), "-I", REF, src, "-o", exe])
        rows = json.loads(subprocess.check_output([exe]).decode())
    fixed, seeded = rows[:10], rows[10:]
    kat = dict(source="reference src/base3d/camera_models.h world2image<double>, compiled in the build container by "
                      "tests/golden/make_world2image_kat.py (g++ -O0); parameter sets of camera_models_test.cc:60-82",
               vectors=[dict(model=r["model"], code=r["code"], Xc=r["Xc"], params=r["params"], uv=r["uv"]) for r in fixed])
    json.dump(kat, open(os.path.join(HERE, "world2image_kat.json"), "w"), indent=1)
    jet = dict(source="reference world2image<T> with T = double and T = forward-mode dual number (3 + K directions: Xc, then the "
                      "K intrinsics); 100 seeded samples per model + the 10 fixed vectors; generator tests/golden/make_world2image_kat.py",
               columns="du / dv = d u / d (x, y, z, k_0 .. k_{K-1})", vectors=rows)
    json.dump(jet, open(os.path.join(HERE, "world2image_jet_kat.json"), "w"), indent=None, separators=(",", ":"))
    print(len(fixed), "fixed +", len(seeded), "seeded vectors written")


if __name__ == "__main__":
    main()
