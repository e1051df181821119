// Per-voxel arithmetic of the DTI scalar maps (reference: src/eval.py:85-116), shared by the HIP kernel
// and by the host-compiled arithmetic check in tests/ (plain C++: g++ sees the qualifiers as empty).
//
// Symmetric 3x3 eigen-decomposition without iteration, accurate to a few ulp of |A| also for (nearly)
// repeated eigenvalues: the ISOLATED extreme eigenvalue comes from the trigonometric solution of the
// characteristic cubic (well conditioned for that root), its eigenvector from the best row cross product
// of (A - lambda I) (rank 2, gap >= sqrt(3) p), and the remaining pair from the exact 2x2 problem in the
// orthogonal complement -- the ill-conditioned acos branch never feeds a close pair.
#pragma once
#include <math.h>
#ifndef __HIPCC__
#define MI355_HD
#else
#define MI355_HD __host__ __device__ __forceinline__
#endif

struct DtiVoxel { double w0, w1, w2; double e[3]; };   // ascending eigenvalues, unit eigenvector of w2 (z >= 0)

MI355_HD void dti_cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

MI355_HD void dti_eigh(double xx, double xy, double xz, double yy, double yz, double zz, DtiVoxel& o) {
  const double p1 = xy * xy + xz * xz + yz * yz;
  if (p1 == 0.0) {
    // diagonal tensor: eigenvalues are the diagonal, eigenvectors the axes (ties -> the later axis)
    int k = 0; double m = xx;
    if (yy >= m) { m = yy; k = 1; }
    if (zz >= m) { m = zz; k = 2; }
    const double o0 = k == 0 ? yy : xx, o1 = k == 2 ? yy : zz;
    o.w2 = m; o.w0 = fmin(o0, o1); o.w1 = fmax(o0, o1);
    o.e[0] = k == 0; o.e[1] = k == 1; o.e[2] = k == 2;
    return;
  }
  const double q = (xx + yy + zz) / 3.0;
  const double bx = xx - q, by = yy - q, bz = zz - q;
  const double p = sqrt((bx * bx + by * by + bz * bz + 2.0 * p1) / 6.0), ip = 1.0 / p;
  const double b00 = bx * ip, b11 = by * ip, b22 = bz * ip, b01 = xy * ip, b02 = xz * ip, b12 = yz * ip;
  double r = 0.5 * (b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02));
  r = fmin(1.0, fmax(-1.0, r));
  const bool top = r >= 0.0;                       // the largest eigenvalue is the isolated one
  const double phi = acos(r) / 3.0;                // [0, pi/3]
  const double lam = q + 2.0 * p * cos(top ? phi : phi + 2.0943951023931954923);
  // eigenvector of lam
  const double r0[3] = {xx - lam, xy, xz}, r1[3] = {xy, yy - lam, yz}, r2[3] = {xz, yz, zz - lam};
  double c0[3], c1[3], c2[3], f[3];
  dti_cross3(r0, r1, c0); dti_cross3(r0, r2, c1); dti_cross3(r1, r2, c2);
  const double n0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
  const double n1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
  const double n2 = c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2];
  double nb = n0; f[0] = c0[0]; f[1] = c0[1]; f[2] = c0[2];
  if (n1 > nb) { nb = n1; f[0] = c1[0]; f[1] = c1[1]; f[2] = c1[2]; }
  if (n2 > nb) { nb = n2; f[0] = c2[0]; f[1] = c2[1]; f[2] = c2[2]; }
  const double inb = 1.0 / sqrt(nb);               // nb >= (sqrt(3) p)^2 * p^2 / 3-ish > 0 since p > 0
  f[0] *= inb; f[1] *= inb; f[2] *= inb;
  // orthonormal basis (u, v) of the complement of f
  double u[3], v[3];
  if (fabs(f[0]) > fabs(f[1])) {
    const double il = 1.0 / sqrt(f[0] * f[0] + f[2] * f[2]);
    u[0] = -f[2] * il; u[1] = 0.0; u[2] = f[0] * il;
  } else {
    const double il = 1.0 / sqrt(f[1] * f[1] + f[2] * f[2]);
    u[0] = 0.0; u[1] = f[2] * il; u[2] = -f[1] * il;
  }
  dti_cross3(f, u, v);
  // 2x2 problem [a b; b c] of A restricted to span(u, v)
  const double au[3] = {xx * u[0] + xy * u[1] + xz * u[2], xy * u[0] + yy * u[1] + yz * u[2], xz * u[0] + yz * u[1] + zz * u[2]};
  const double av[3] = {xx * v[0] + xy * v[1] + xz * v[2], xy * v[0] + yy * v[1] + yz * v[2], xz * v[0] + yz * v[1] + zz * v[2]};
  const double a = u[0] * au[0] + u[1] * au[1] + u[2] * au[2];
  const double b = v[0] * au[0] + v[1] * au[1] + v[2] * au[2];
  const double c = v[0] * av[0] + v[1] * av[1] + v[2] * av[2];
  const double mean = 0.5 * (a + c), diff = 0.5 * (a - c), rad = sqrt(diff * diff + b * b);
  const double hi = mean + rad, lo = mean - rad;
  if (top) {
    o.w2 = lam; o.w1 = hi; o.w0 = lo;
    o.e[0] = f[0]; o.e[1] = f[1]; o.e[2] = f[2];
  } else {
    o.w0 = lam; o.w1 = lo; o.w2 = hi;
    // eigenvector of `hi` in the (u, v) basis, from the row without cancellation; rad == 0 -> any (take u)
    double cx = 1.0, cy = 0.0;
    if (rad > 0.0) {
      if (diff >= 0.0) { cx = diff + rad; cy = b; } else { cx = b; cy = rad - diff; }
      const double il = 1.0 / sqrt(cx * cx + cy * cy);
      cx *= il; cy *= il;
    }
    o.e[0] = cx * u[0] + cy * v[0]; o.e[1] = cx * u[1] + cy * v[1]; o.e[2] = cx * u[2] + cy * v[2];
  }
}

// out: fa, md, ad, rd, azimuth, inclination, rgb[3].  `f32_angles`: angle functions in f32 (f32 outputs).
MI355_HD void dti_voxel_maps(const double* d, bool f32_angles, double* out) {
  DtiVoxel o;
  dti_eigh(d[0], d[1], d[2], d[3], d[4], d[5], o);
  double* e = o.e;
  // eigenvector sign is arbitrary (LAPACK's choice in the reference): directions are axial, take z >= 0
  if (e[2] < 0.0 || (e[2] == 0.0 && (e[1] < 0.0 || (e[1] == 0.0 && e[0] < 0.0)))) { e[0] = -e[0]; e[1] = -e[1]; e[2] = -e[2]; }
  const double w0 = o.w0, w1 = o.w1, w2 = o.w2;
  const double md = (w0 + w1 + w2) / 3.0;
  const double u0 = w0 - md, u1 = w1 - md, u2 = w2 - md;
  const double var = sqrt(u0 * u0 + u1 * u1 + u2 * u2);
  const double nrm = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
  const double fa = 1.2247448713915890491 * var / nrm;      // sqrt(1.5); 0/0 -> NaN like numpy
  const double deg = 57.295779513082320877;
  double az, inc;
  if (f32_angles) {
    // error ~1e-5 degree, the rounding of the stored f32 value
    az = deg * (double)atan2f((float)e[1], (float)e[0]);
    // e is unit and e[2] >= 0: acos via atan2 keeps full relative accuracy near the pole
    inc = deg * (double)atan2f((float)sqrt(e[0] * e[0] + e[1] * e[1]), (float)e[2]);
  } else {
    az = deg * atan2(e[1], e[0]);
    const double rlen = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    inc = deg * acos(fmin(1.0, fmax(-1.0, e[2] / rlen)));
  }
  if (az > 180.0) az -= 360.0;
  out[0] = fa; out[1] = md; out[2] = w2; out[3] = (w0 + w1) * 0.5; out[4] = az; out[5] = inc;
  out[6] = fa * fabs(e[0]); out[7] = fa * fabs(e[1]); out[8] = fa * fabs(e[2]);
}
