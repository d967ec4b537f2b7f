// Host build (g++) of mavmap_amd/csrc/ba_math.h — TEST HARNESS ONLY.
// Lets the CPU test-suite check the kernel bodies against the oracle without a GPU.
#include "../../mavmap_amd/csrc/ba_math.h"
extern "C" {
void hm_obs_jacobian(int model, const double* pose, const double* cam, const double* X,
                     const double* uv, double* r, double* Jc, double* Jp, double* Jk) {
  double rec[9];
  mavba::cam_prepare(pose, rec);
  mavba::obs_jacobian(model, rec, cam, X, uv[0], uv[1], r, Jc, Jp, Jk);
}
void hm_obs_residual(int model, const double* pose, const double* cam, const double* X,
                     const double* uv, double* r) {
  double rec[9];
  mavba::cam_prepare(pose, rec);
  mavba::obs_residual(model, rec, cam, X, uv[0], uv[1], r);
}
void hm_obs_backsub_term(int model, const double* pose, const double* cam, const double* X, const double* uv,
                         const double* dc, const double* dk, double* r, double* t) {
  double rec[9];
  mavba::cam_prepare(pose, rec);
  mavba::obs_backsub_term(model, rec, cam, X, uv[0], uv[1], dc, dk, r, t);
}
void hm_rot_prior(const double* w, const double* w0, double weight, double* res, double* jac) {
  double R0[9];
  mavba::rot_matrix_colmajor(w0, R0);
  mavba::rot_prior_eval(w, R0, weight, *res, jac);
}
void hm_rot_matrix(const double* w, double* R) { mavba::rot_matrix_colmajor(w, R); }
int hm_chol3_inv(const double* C, double* Gi) { return mavba::chol3_inv(C, Gi) ? 1 : 0; }
void hm_cauchy(double s, double a, double* w, double* half_rho) {
  mavba::cauchy_weight(s, a * a, 1.0 / (a * a), *w, *half_rho);
}
// the device build's log series (log_obs) on the host: frexp supplies the parts the hardware instructions give the kernel
double hm_log_series(double x) { int e; const double m = std::frexp(x, &e); return mavba::log_from_parts(m, e); }
}
