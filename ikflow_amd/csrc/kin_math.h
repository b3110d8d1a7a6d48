// kin_math.h - the per-row arithmetic of the kinematics kernels (chain walk, quaternion, pose error, Jacobian, the LM step in both of its
// arithmetics): everything kin_kernels.hip runs per thread, WITHOUT anything of the HIP runtime in it, so that the same source also compiles
// with g++ - tests/test_kin_math_host.py runs it on the CPU against the oracle (how round 6 found that fused multiply-adds in the fp32 LU put
// poses next to a singularity 6 x outside the reference's noise: the device code on the host did not show it, the device did).
// Test infrastructure compiles this header; the product only ever runs it on the GPU (the library has no CPU path).
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/ikflow_amd.h"

#if defined(__HIPCC__)
#define IKF_HD __device__ __forceinline__
#else
#define IKF_HD inline
#endif

namespace ikf {

// the robot as the kernels see it: actuated joints with the fixed transforms in front of them folded in (engine.fold_chain), tool, limits
struct Chain {
  int ndof;
  ikf_joint joints[IKF_MAX_DOF];
  float tool[12];
  float lo[IKF_MAX_DOF];
  float hi[IKF_MAX_DOF];
};

template <typename T>
IKF_HD void compose(T R[9], T p[3], const float* __restrict__ pre) {
  // (R,p) <- (R,p) * (Rf,pf),  pre = 3x4 row-major [Rf | pf]
  T Rn[9], pn[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      Rn[3 * r + c] = R[3 * r + 0] * (T)pre[0 * 4 + c] + R[3 * r + 1] * (T)pre[1 * 4 + c] + R[3 * r + 2] * (T)pre[2 * 4 + c];
    pn[r] = R[3 * r + 0] * (T)pre[3] + R[3 * r + 1] * (T)pre[7] + R[3 * r + 2] * (T)pre[11] + p[r];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = Rn[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = pn[i];
}

IKF_HD void sincos_t(float a, float* s, float* c) { sincosf(a, s, c); }
IKF_HD void sincos_t(double a, double* s, double* c) { sincos(a, s, c); }

template <typename T>
IKF_HD void apply_joint(T R[9], T p[3], int kind, const float* __restrict__ axis, T qv) {
  const T x = (T)axis[0], y = (T)axis[1], z = (T)axis[2];
  if (kind == 1) {
    T s, c;
    sincos_t(qv, &s, &c);
    const T t = (T)1 - c;
    T M[9] = {t * x * x + c,     t * x * y - s * z, t * x * z + s * y,
              t * x * y + s * z, t * y * y + c,     t * y * z - s * x,
              t * x * z - s * y, t * y * z + s * x, t * z * z + c};
    T Rn[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
        Rn[3 * r + cc] = R[3 * r + 0] * M[cc] + R[3 * r + 1] * M[3 + cc] + R[3 * r + 2] * M[6 + cc];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Rn[i];
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r) p[r] += (R[3 * r + 0] * x + R[3 * r + 1] * y + R[3 * r + 2] * z) * qv;
  }
}

// Full chain walk. If RECORD, also returns each joint's world axis and world origin (taken after the joint's fixed
// pre-transform, before its own motion) for the geometric Jacobian.
template <typename T, int NDOF, bool RECORD>
IKF_HD void fk_walk(const Chain* __restrict__ ch, const T q[NDOF], T R[9], T p[3],
                                        T axis_w[][3], T org_w[][3]) {
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? (T)1 : (T)0;
  p[0] = p[1] = p[2] = (T)0;
#pragma unroll
  for (int j = 0; j < NDOF; ++j) {
    compose<T>(R, p, ch->joints[j].pre);
    if (RECORD) {
      const float* ax = ch->joints[j].axis;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        axis_w[j][r] = R[3 * r + 0] * (T)ax[0] + R[3 * r + 1] * (T)ax[1] + R[3 * r + 2] * (T)ax[2];
        org_w[j][r] = p[r];
      }
    }
    apply_joint<T>(R, p, ch->joints[j].kind, ch->joints[j].axis, q[j]);
  }
  compose<T>(R, p, ch->tool);
}

IKF_HD float sqrt_t(float a) { return sqrtf(a); }
IKF_HD double sqrt_t(double a) { return sqrt(a); }

// rotation matrix -> (w,x,y,z): candidate built from the largest of |w|,|x|,|y|,|z| (that component positive)
template <typename T>
IKF_HD void mat_to_quat(const T R[9], T qo[4]) {
  const T m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6], m21 = R[7], m22 = R[8];
  T qa[4];
  qa[0] = sqrt_t(fmax((T)0, (T)1 + m00 + m11 + m22));
  qa[1] = sqrt_t(fmax((T)0, (T)1 + m00 - m11 - m22));
  qa[2] = sqrt_t(fmax((T)0, (T)1 - m00 + m11 - m22));
  qa[3] = sqrt_t(fmax((T)0, (T)1 - m00 - m11 + m22));
  int best = 0;
  T bv = qa[0];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (qa[i] > bv) { bv = qa[i]; best = i; }
  const T den = (T)2 * fmax(bv, (T)0.1);
  T c0, c1, c2, c3;
  if (best == 0)      { c0 = qa[0] * qa[0]; c1 = m21 - m12;       c2 = m02 - m20;       c3 = m10 - m01; }
  else if (best == 1) { c0 = m21 - m12;     c1 = qa[1] * qa[1];   c2 = m10 + m01;       c3 = m02 + m20; }
  else if (best == 2) { c0 = m02 - m20;     c1 = m10 + m01;       c2 = qa[2] * qa[2];   c3 = m12 + m21; }
  else                { c0 = m10 - m01;     c1 = m20 + m02;       c2 = m21 + m12;       c3 = qa[3] * qa[3]; }
  qo[0] = c0 / den; qo[1] = c1 / den; qo[2] = c2 / den; qo[3] = c3 / den;
}

IKF_HD float geodesic_f32(const float* qa, const float* qb) {
  const float lo = (float)(-1.0 + 1e-7), hi = (float)(1.0 - 1e-7);
  float dot = qa[0] * qb[0] + qa[1] * qb[1] + qa[2] * qb[2] + qa[3] * qb[3];
  dot = fminf(fmaxf(dot, lo), hi);
  const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
  float d = 2.0f * acosf(dot);
  float m = fmodf(d + PI_F, TWO_PI_F);
  if (m < 0.f) m += TWO_PI_F;
  return fabsf(m - PI_F);
}

template <int NDOF>
IKF_HD void load_q(const float* __restrict__ q, long long row, float out[NDOF]) {
#pragma unroll
  for (int j = 0; j < NDOF; ++j) out[j] = q[(size_t)row * NDOF + j];
}

template <int NDOF>
IKF_HD void fk_pose_f32(const Chain* __restrict__ ch, const float qv[NDOF], float pose[7]) {
  float R[9], p[3];
  fk_walk<float, NDOF, false>(ch, qv, R, p, nullptr, nullptr);
  float qq[4];
  mat_to_quat<float>(R, qq);
  pose[0] = p[0]; pose[1] = p[1]; pose[2] = p[2];
  pose[3] = qq[0]; pose[4] = qq[1]; pose[5] = qq[2]; pose[6] = qq[3];
}

template <int NDOF>
IKF_HD void pose_error_f32(const Chain* __restrict__ ch, const float qv[NDOF],
                                               const float* __restrict__ tgt, float* pos_err, float* rot_err) {
  float pose[7];
  fk_pose_f32<NDOF>(ch, qv, pose);
  const float dx = pose[0] - tgt[0], dy = pose[1] - tgt[1], dz = pose[2] - tgt[2];
  *pos_err = sqrtf(dx * dx + dy * dy + dz * dz);
  const float tq[4] = {tgt[3], tgt[4], tgt[5], tgt[6]};
  *rot_err = geodesic_f32(tq, pose + 3);
}

IKF_HD float atan2_t(float y, float x) { return atan2f(y, x); }
IKF_HD double atan2_t(double y, double x) { return atan2(y, x); }
IKF_HD float asin_t(float a) { return asinf(a); }
IKF_HD double asin_t(double a) { return asin(a); }

// Solve A x = g in place (x -> g), A symmetric positive definite, lower triangle filled: Cholesky A = L L^T (the fp64 mode)
template <int NDOF>
IKF_HD void solve_cholesky(double A[NDOF][NDOF], double g[NDOF]) {
#pragma unroll
  for (int c = 0; c < NDOF; ++c) {
    double dsum = A[c][c];
#pragma unroll
    for (int k = 0; k < c; ++k) dsum -= A[c][k] * A[c][k];
    const double lcc = sqrt(dsum);
    A[c][c] = lcc;
    const double inv = 1.0 / lcc;
#pragma unroll
    for (int r = c + 1; r < NDOF; ++r) {
      double v = A[r][c];
#pragma unroll
      for (int k = 0; k < c; ++k) v -= A[r][k] * A[c][k];
      A[r][c] = v * inv;
    }
  }
#pragma unroll
  for (int r = 0; r < NDOF; ++r) {
    double v = g[r];
#pragma unroll
    for (int k = 0; k < r; ++k) v -= A[r][k] * g[k];
    g[r] = v / A[r][r];
  }
#pragma unroll
  for (int r = NDOF - 1; r >= 0; --r) {
    double v = g[r];
#pragma unroll
    for (int k = r + 1; k < NDOF; ++k) v -= A[k][r] * g[k];
    g[r] = v / A[r][r];
  }
}

// Solve A x = g in place (x -> g), A full: LU with partial pivoting, then the two triangular solves - what torch.linalg.solve does on the
// reference's fp32 tensors (LAPACK sgesv = sgetrf + sgetrs; ikflow_solver.py:205,208 -> jrl).  Everything stays in registers: the pivot
// row is found by an unrolled compare and the swap is an unrolled select, so no index is a run-time value.
// Every operation of the elimination is rounded on its own (no fused multiply-adds), as in the reference routine: with the compiler's
// contraction the poses next to a singularity (TWO eigenvalues of J^T J + 1e-4 I near 1e-4) came out up to 9.5 x cond x 2^-24 x |dq| from the fp64
// step - 6 x outside what the oracle's sgesv, and this code without contraction, leave (r06, tools/lm_precision_report.py; the median and p99 were
// the same either way).  Cost: nothing measurable (a 7 x 7 solve per row).
template <int NDOF>
IKF_HD void solve_lu_pivot(float A[NDOF][NDOF], float g[NDOF]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int c = 0; c < NDOF; ++c) {
    int piv = c;
    float best = fabsf(A[c][c]);
#pragma unroll
    for (int r = c + 1; r < NDOF; ++r) {
      const float v = fabsf(A[r][c]);
      if (v > best) { best = v; piv = r; }   // (first maximum wins, as isamax)
    }
#pragma unroll
    for (int r = c + 1; r < NDOF; ++r) {
      const bool sw = piv == r;
#pragma unroll
      for (int k = 0; k < NDOF; ++k) {
        const float a = A[c][k], b = A[r][k];
        A[c][k] = sw ? b : a;
        A[r][k] = sw ? a : b;
      }
      const float ga = g[c], gb = g[r];
      g[c] = sw ? gb : ga;
      g[r] = sw ? ga : gb;
    }
    const float inv = 1.0f / A[c][c];
#pragma unroll
    for (int r = c + 1; r < NDOF; ++r) {
      const float l = A[r][c] * inv;
#pragma unroll
      for (int k = c + 1; k < NDOF; ++k) A[r][k] -= l * A[c][k];
      g[r] -= l * g[c];   // (forward substitution with the unit lower factor, applied as the factor is formed)
    }
  }
#pragma unroll
  for (int r = NDOF - 1; r >= 0; --r) {
    float v = g[r];
#pragma unroll
    for (int k = r + 1; k < NDOF; ++k) v -= A[r][k] * g[k];
    g[r] = v / A[r][r];
  }
}

// One damped least-squares step: q <- clamp(q + (J^T J + 1e-4 I)^-1 J^T e).
//   T = double (default, lm_precision 1): chain walk, Jacobian, normal equations and a Cholesky solve in fp64, q rounded to fp32 at the end;
//   T = float  (lm_precision 0): the reference's own arithmetic - every quantity fp32 (ikflow/config.py:8), LU with partial pivoting.
template <int NDOF, typename T>
IKF_HD void lm_step_row(const Chain* __restrict__ ch, const float* __restrict__ tgt,
                                            float qv[NDOF]) {
  T qd[NDOF];
#pragma unroll
  for (int j = 0; j < NDOF; ++j) qd[j] = (T)qv[j];
  T R[9], p[3], axw[NDOF][3], orw[NDOF][3];
  fk_walk<T, NDOF, true>(ch, qd, R, p, axw, orw);
  T qc[4];
  mat_to_quat<T>(R, qc);
  // rotation error quaternion = q_target * conj(q_current), as roll/pitch/yaw
  const T w1 = tgt[3], x1 = tgt[4], y1 = tgt[5], z1 = tgt[6];
  const T w2 = qc[0], x2 = -qc[1], y2 = -qc[2], z2 = -qc[3];
  const T ew = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
  const T ex = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
  const T ey = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
  const T ez = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
  T e[6];
  e[0] = atan2_t((T)2 * (ew * ex + ey * ez), (T)1 - (T)2 * (ex * ex + ey * ey));
  e[1] = asin_t(fmin(fmax((T)2 * (ew * ey - ez * ex), (T)-1), (T)1));
  e[2] = atan2_t((T)2 * (ew * ez + ex * ey), (T)1 - (T)2 * (ey * ey + ez * ez));
  e[3] = (T)tgt[0] - p[0];
  e[4] = (T)tgt[1] - p[1];
  e[5] = (T)tgt[2] - p[2];
  // Jacobian columns: revolute [axis; axis x (p_ee - origin)], prismatic [0; axis]
  T J[6][NDOF];
#pragma unroll
  for (int j = 0; j < NDOF; ++j) {
    if (ch->joints[j].kind == 1) {
      const T rx = p[0] - orw[j][0], ry = p[1] - orw[j][1], rz = p[2] - orw[j][2];
      J[0][j] = axw[j][0]; J[1][j] = axw[j][1]; J[2][j] = axw[j][2];
      J[3][j] = axw[j][1] * rz - axw[j][2] * ry;
      J[4][j] = axw[j][2] * rx - axw[j][0] * rz;
      J[5][j] = axw[j][0] * ry - axw[j][1] * rx;
    } else {
      J[0][j] = J[1][j] = J[2][j] = (T)0;
      J[3][j] = axw[j][0]; J[4][j] = axw[j][1]; J[5][j] = axw[j][2];
    }
  }
  T A[NDOF][NDOF], g[NDOF];
#pragma unroll
  for (int a = 0; a < NDOF; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      T sacc = (T)0;
#pragma unroll
      for (int r = 0; r < 6; ++r) sacc += J[r][a] * J[r][b];
      A[a][b] = sacc + (a == b ? (T)1e-4 : (T)0);
      A[b][a] = A[a][b];
    }
    T gs = (T)0;
#pragma unroll
    for (int r = 0; r < 6; ++r) gs += J[r][a] * e[r];
    g[a] = gs;
  }
  if constexpr (sizeof(T) == 8) solve_cholesky<NDOF>(A, g);
  else solve_lu_pivot<NDOF>(A, g);
#pragma unroll
  for (int j = 0; j < NDOF; ++j) {
    const float qn = (float)(qd[j] + g[j]);
    qv[j] = fminf(fmaxf(qn, ch->lo[j]), ch->hi[j]);
  }
}

}  // namespace ikf
