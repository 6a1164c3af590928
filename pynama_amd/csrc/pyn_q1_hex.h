// Q1 hexahedron building blocks shared by the patch-plan kernels (pyn_assemble_tiled.hip) and the plan-free lattice
// kernels (pyn_assemble_lattice.hip): quadrature-point update, parallelepiped test and shortcut, closed-form integer
// reference matrices, Jacobian helpers.  Reference formulas: src/elements/spectral.py:117-131.
#pragma once
#include "pyn_internal.h"

namespace {

constexpr int TILE_THREADS = 256;

struct TileArgs {
  const int32_t* conn;
  const double* xyz;
  const int32_t* rowptr;
  const int32_t* colidx;
  const uint8_t* bcmask;  // per node (scalar forms), may be null
  const uint8_t* colbc;   // per CSR entry: column node imposed (null iff bcmask null)
  const int32_t* p_rowptr;
  const int32_t* p_rows;
  const int32_t* p_eptr;
  const int32_t* p_elem;
  const uint4* rowslot4;
  const uint4* kmap4;
  int64_t npe;
  int n_patch;
  int maxlen;           // max CSR row length (27)
  int maxrows;          // max rows per patch (LDS layout)
  const double* w;      // full rule [8]
  const double* hrs;    // [8][3][8]  reference gradients of the nodal basis at the Gauss points
  const double* hcoo;   // [8][3][8]  reference gradients of the geometry (corner) basis
  const double* aff;    // [6][36] affine reference matrices + [4][8] monomial signs (null: shortcut off)
  int lean = 0;         // the uploaded tables are the standard 2x2x2 Gauss tables: general elements take q1_laplace_lean36
  double* A;            // values for free columns
  double* Arhs;         // -values for imposed columns (may be null)
};

// One Gauss point: J = hcoo.X, Ji = J^-1, c = w detJ, G = Ji.hrs, L += c G^T G (upper triangle).
__device__ __forceinline__ void gauss_point(const TileArgs& T, const int G, const double (&X)[8][3], double (&L)[36]) {
  const double* __restrict__ hc = T.hcoo + G * 24;
  const double* __restrict__ hr = T.hrs + G * 24;
  double J[3][3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) s = fma(hc[d * 8 + c], X[c][x], s);
      J[d][x] = s;
    }
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
  const double r = 1.0 / det;
  double Ji[3][3];
  Ji[0][0] = c00 * r;
  Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
  Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
  Ji[1][0] = c01 * r;
  Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
  Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
  Ji[2][0] = c02 * r;
  Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
  Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
  const double cw = T.w[G] * det;
  double Gm[3][8];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      double s = Ji[d][0] * hr[a];
      s = fma(Ji[d][1], hr[8 + a], s);
      s = fma(Ji[d][2], hr[16 + a], s);
      Gm[d][a] = s;
    }
  int idx = 0;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const double g0 = cw * Gm[0][a], g1 = cw * Gm[1][a], g2 = cw * Gm[2][a];
#pragma unroll
    for (int b = a; b < 8; ++b) {
      double s = L[idx];
      s = fma(g0, Gm[0][b], s);
      s = fma(g1, Gm[1][b], s);
      s = fma(g2, Gm[2][b], s);
      L[idx++] = s;
    }
  }
}

// Affine shortcut: for a parallelepiped J is constant and the 2x2x2 rule integrates the (quadratic)
// integrand exactly, so L_ab = detJ * sum_{r<=s} Q_rs T_rs[ab] with Q = J^-T J^-1 -- ~350 instead of
// ~2500 FP64 operations.  `affine` is decided per element from the non-affine trilinear modes.
__device__ __forceinline__ bool element_is_affine(const TileArgs& T, const double (&X)[8][3]) {
  const double* __restrict__ sg = T.aff + 216;
  double na = 0.0, h2 = 0.0;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      double c = 0.0;
#pragma unroll
      for (int a = 0; a < 8; ++a) c = fma(sg[m * 8 + a], X[a][x], c);
      na = fma(c, c, na);
    }
#pragma unroll
  for (int x = 0; x < 3; ++x) {  // squared edge scale: (x_6 - x_0) carries all three affine modes
    const double d = X[6][x] - X[0][x];
    h2 = fma(d, d, h2);
  }
  return na <= 1e-25 * h2;  // non-affine modes below ~3e-13 of the element size: coordinate round-off
}

__device__ __forceinline__ void affine_laplace(const TileArgs& T, const double (&X)[8][3], double (&L)[36]) {
  const double* __restrict__ hc = T.hcoo;  // any Gauss point: J is constant
  double J[3][3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) s = fma(hc[d * 8 + c], X[c][x], s);
      J[d][x] = s;
    }
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
  const double r = 1.0 / det;
  double Ji[3][3];  // Ji[x][d]: physical axis x, reference axis d   (G = Ji . hr)
  Ji[0][0] = c00 * r;
  Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
  Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
  Ji[1][0] = c01 * r;
  Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
  Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
  Ji[2][0] = c02 * r;
  Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
  Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
  double Q[6];
  const int RS[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {0, 2}, {1, 2}};
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const int a = RS[t][0], b = RS[t][1];
    Q[t] = det * (Ji[0][a] * Ji[0][b] + Ji[1][a] * Ji[1][b] + Ji[2][a] * Ji[2][b]);
  }
  const double* __restrict__ Tm = T.aff;
#pragma unroll
  for (int i = 0; i < 36; ++i) {
    double s = Q[0] * Tm[i];
#pragma unroll
    for (int t = 1; t < 6; ++t) s = fma(Q[t], Tm[t * 36 + i], s);
    L[i] = s;
  }
}

__device__ inline int tri(int a, int b) {  // index of (min,max) in the packed upper triangle of 8x8
  int i = a < b ? a : b, j = a < b ? b : a;
  return i * 8 - (i * (i - 1)) / 2 + (j - i);
}

// 72 * T_rs[a][b] of the trilinear hexahedron in the reference's corner order, in closed form from the corner
// signs s_d(a) (tensor product of the 1-D integrals  int N_i N_j = (3 + s_i s_j)/6,  int N_i' N_j' = s_i s_j/2,
// int N_i' N_j = s_i/2):  rr: s_r(a)s_r(b)(3+s_p s_p)(3+s_q s_q);  rs: 3(3+s_u s_u)(s_r(a)s_s(b)+s_s(a)s_r(b)).
// pyn_elem_tables_set checks the uploaded tables against it (lat_aff_standard) before the lean path is used.
__host__ __device__ constexpr int q1_aff_int(int t, int a, int b) {
  constexpr int SG[3][8] = {{-1, -1, 1, 1, -1, 1, 1, -1}, {-1, 1, 1, -1, -1, -1, 1, 1}, {-1, -1, -1, -1, 1, 1, 1, 1}};
  constexpr int RS[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {0, 2}, {1, 2}};
  const int r = RS[t][0], s2 = RS[t][1];
  if (r == s2) {
    const int p = (r + 1) % 3, q = (r + 2) % 3;
    return SG[r][a] * SG[r][b] * (3 + SG[p][a] * SG[p][b]) * (3 + SG[q][a] * SG[q][b]);
  }
  const int u = 3 - r - s2;
  return 3 * (3 + SG[u][a] * SG[u][b]) * (SG[r][a] * SG[s2][b] + SG[s2][a] * SG[r][b]);
}

// 72 * int N_a d_d N_b over the reference cube = s_d(b) (3 + s_e(a)s_e(b)) (3 + s_f(a)s_f(b)), e, f the other axes
// (pyn_elem_tables_set checks sum_g w_g H_g[a] Hrs_g[d][b] against it before the affine Rw path is used)
__host__ __device__ constexpr int q1_mix_int(int d, int a, int b) {
  constexpr int SG[3][8] = {{-1, -1, 1, 1, -1, 1, 1, -1}, {-1, 1, 1, -1, -1, -1, 1, 1}, {-1, -1, -1, -1, 1, 1, 1, 1}};
  const int e = (d + 1) % 3, f = (d + 2) % 3;
  return SG[d][b] * (3 + SG[e][a] * SG[e][b]) * (3 + SG[f][a] * SG[f][b]);
}

// geometry at one point of a rule: Ji = (hc . X)^-1, returns detJ; G[d][a] = sum_r Ji[d][r] hr[r][a]
__device__ __forceinline__ double point_gradients(const double* __restrict__ hc, const double* __restrict__ hr,
                                                  const double (&X)[8][3], double (&G)[3][8]) {
  double J[3][3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) s = fma(hc[d * 8 + c], X[c][x], s);
      J[d][x] = s;
    }
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
  const double r = 1.0 / det;
  double Ji[3][3];
  Ji[0][0] = c00 * r;
  Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
  Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
  Ji[1][0] = c01 * r;
  Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
  Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
  Ji[2][0] = c02 * r;
  Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
  Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      double s = Ji[d][0] * hr[a];
      s = fma(Ji[d][1], hr[8 + a], s);
      s = fma(Ji[d][2], hr[16 + a], s);
      G[d][a] = s;
    }
  return det;
}

// Ji[x][d] = (hc . X)^-1 (physical axis x, reference axis d), returns detJ
__device__ __forceinline__ double jacobian_inverse(const double* __restrict__ hc, const double (&X)[8][3], double (&Ji)[3][3]) {
  double J[3][3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) s = fma(hc[d * 8 + c], X[c][x], s);
      J[d][x] = s;
    }
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
  const double r = 1.0 / det;
  Ji[0][0] = c00 * r;
  Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
  Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
  Ji[1][0] = c01 * r;
  Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
  Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
  Ji[2][0] = c02 * r;
  Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
  Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
  return det;
}

// ---- lean closed-form element Laplacian (general geometry, standard 2x2x2 Gauss tables: pyn_q1_gauss_tables_standard) ----
constexpr double Q1_GP = 0.57735026918962576451;   // 1/sqrt(3): the 2-point Gauss abscissa (utilities.py:43-61, N = 2)
constexpr int Q1_NODE[2][2][2] = {{{0, 4}, {1, 7}}, {{3, 5}, {2, 6}}};   // [i][j][k] -> local node

// index of pair (a < b) in the packed strict upper triangle of 8 x 8
__host__ __device__ constexpr int q1_off(int a, int b) { return a * 7 - (a * (a - 1)) / 2 + (b - a - 1); }

// unnormalised Haar coefficients of the corner coordinates: C[bz][by][bx][c] = sum_a sx^bx sy^by sz^bz X_a[c]
// (8 x the coefficient of xi^bx eta^by zeta^bz in the trilinear map).  P[k][j][i][c] = corner (i, j, k).
__device__ __forceinline__ void q1_haar_coeffs(const double (&P)[2][2][2][3], double (&C)[2][2][2][3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double a[2][2][2], b[2][2][2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        a[k][j][0] = P[k][j][1][c] + P[k][j][0][c];
        a[k][j][1] = P[k][j][1][c] - P[k][j][0][c];
      }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        b[k][0][i] = a[k][1][i] + a[k][0][i];
        b[k][1][i] = a[k][1][i] - a[k][0][i];
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        C[0][j][i][c] = b[1][j][i] + b[0][j][i];
        C[1][j][i][c] = b[1][j][i] - b[0][j][i];
      }
  }
}

// reciprocal by v_rcp_f64 + two Newton steps (the determinant of a valid element is far from the ends of the range)
__device__ __forceinline__ double q1_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  return fma(r, e, r);
}

// Geometry at the Gauss point (SX, SY, SZ)/sqrt(3), everything unnormalised (J' = 8 J):  rows r_d = d x / d xi_d from
// the Haar coefficients, A_d = r_{d+1} x r_{d+2} (the columns of adj J'), det' = r_0 . A_0 = 512 detJ.  Physical
// gradients:  G[x][a] = (sum_d A_d[x] h_d[a]) / det'  with  h_d[a] = s_d(a) prod_{e != d} (1 + s_e(a) xi_e)  = 8 Hrs.
template <int SX, int SY, int SZ>
__device__ __forceinline__ double q1_point_adj(const double (&C)[2][2][2][3], double (&A)[3][3]) {
  constexpr double xi = SX * Q1_GP, eta = SY * Q1_GP, zeta = SZ * Q1_GP;
  double r0[3], r1[3], r2[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double m = fma(xi, C[1][1][1][c], C[1][1][0][c]);   // shared by the eta- and zeta-rows
    r0[c] = fma(zeta, fma(eta, C[1][1][1][c], C[1][0][1][c]), fma(eta, C[0][1][1][c], C[0][0][1][c]));
    r1[c] = fma(zeta, m, fma(xi, C[0][1][1][c], C[0][1][0][c]));
    r2[c] = fma(eta, m, fma(xi, C[1][0][1][c], C[1][0][0][c]));
  }
  A[0][0] = r1[1] * r2[2] - r1[2] * r2[1];
  A[0][1] = r1[2] * r2[0] - r1[0] * r2[2];
  A[0][2] = r1[0] * r2[1] - r1[1] * r2[0];
  A[1][0] = r2[1] * r0[2] - r2[2] * r0[1];
  A[1][1] = r2[2] * r0[0] - r2[0] * r0[2];
  A[1][2] = r2[0] * r0[1] - r2[1] * r0[0];
  A[2][0] = r0[1] * r1[2] - r0[2] * r1[1];
  A[2][1] = r0[2] * r1[0] - r0[0] * r1[2];
  A[2][2] = r0[0] * r1[1] - r0[1] * r1[0];
  return r0[0] * A[0][0] + r0[1] * A[0][1] + r0[2] * A[0][2];
}

// h_d[a] at the Gauss point (SX, SY, SZ)/sqrt(3): compile-time constants
template <int SX, int SY, int SZ>
__host__ __device__ constexpr double q1_h(int d, int i, int j, int k) {
  const double px = 1.0 + (2 * i - 1) * SX * Q1_GP, py = 1.0 + (2 * j - 1) * SY * Q1_GP, pz = 1.0 + (2 * k - 1) * SZ * Q1_GP;
  return d == 0 ? (2 * i - 1) * py * pz : (d == 1 ? (2 * j - 1) * px * pz : (2 * k - 1) * px * py);
}

// one Gauss point of the scalar Laplacian:  L_ab += (w / 512) / det' * sum_x g_x[a] g_x[b],  g_x = sum_d A_d[x] h_d
// (off-diagonal pairs only); ws = w_g / 512
template <int SX, int SY, int SZ>
__device__ __forceinline__ void q1_laplace_point(const double (&C)[2][2][2][3], const double ws, double (&L)[28]) {
  double A[3][3];
  const double det = q1_point_adj<SX, SY, SZ>(C, A);
  const double s = ws * q1_rcp(det);
#pragma unroll
  for (int x = 0; x < 3; ++x) {
    double g[8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k)
          g[Q1_NODE[i][j][k]] = fma(A[2][x], q1_h<SX, SY, SZ>(2, i, j, k),
                                    fma(A[1][x], q1_h<SX, SY, SZ>(1, i, j, k), A[0][x] * q1_h<SX, SY, SZ>(0, i, j, k)));
#pragma unroll
    for (int a = 0; a < 7; ++a) {
      const double t = s * g[a];
#pragma unroll
      for (int b = a + 1; b < 8; ++b) L[q1_off(a, b)] = fma(t, g[b], L[q1_off(a, b)]);
    }
  }
}

// the 28 off-diagonal entries of the element Laplacian (spectral.py:117-131 restricted to one component)
__device__ __forceinline__ void q1_laplace_lean(const double (&P)[2][2][2][3], const double ws, double (&L)[28]) {
  double C[2][2][2][3];
  q1_haar_coeffs(P, C);
#pragma unroll
  for (int i = 0; i < 28; ++i) L[i] = 0.0;
  q1_laplace_point<-1, -1, -1>(C, ws, L);
  q1_laplace_point<+1, -1, -1>(C, ws, L);
  q1_laplace_point<-1, +1, -1>(C, ws, L);
  q1_laplace_point<+1, +1, -1>(C, ws, L);
  q1_laplace_point<-1, -1, +1>(C, ws, L);
  q1_laplace_point<+1, -1, +1>(C, ws, L);
  q1_laplace_point<-1, +1, +1>(C, ws, L);
  q1_laplace_point<+1, +1, +1>(C, ws, L);
}

// ---- the same point routines with the Gauss point as a RUN-TIME index: a rolled loop over the eight points keeps the register
// demand at one point's worth (the fully unrolled form lets the scheduler interleave the points: 270+ VGPRs).  The constants of a
// point (its coordinates, h_d[a], N'_a / 4096) come from a __constant__ table through scalar loads.
struct Q1PointTab {
  double xi[3];
  double h[3][8];     // h_d[a] = s_d(a) prod_{e != d} (1 + s_e(a) xi_e), reference node order
  double n[8];        // prod_d (1 + s_d(a) xi_d) / 4096
};

__host__ __device__ constexpr Q1PointTab q1_point_tab(int G) {
  constexpr int CX[8] = {0, 0, 1, 1, 0, 1, 1, 0};
  constexpr int CY[8] = {0, 1, 1, 0, 0, 0, 1, 1};
  constexpr int CZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  Q1PointTab t{};
  const double sx = ((G & 1) * 2 - 1) * Q1_GP, sy = (((G >> 1) & 1) * 2 - 1) * Q1_GP, sz = (((G >> 2) & 1) * 2 - 1) * Q1_GP;
  t.xi[0] = sx;
  t.xi[1] = sy;
  t.xi[2] = sz;
  for (int a = 0; a < 8; ++a) {
    const double px = 1.0 + (2 * CX[a] - 1) * sx, py = 1.0 + (2 * CY[a] - 1) * sy, pz = 1.0 + (2 * CZ[a] - 1) * sz;
    t.h[0][a] = (2 * CX[a] - 1) * py * pz;
    t.h[1][a] = (2 * CY[a] - 1) * px * pz;
    t.h[2][a] = (2 * CZ[a] - 1) * px * py;
    t.n[a] = px * py * pz * (1.0 / 4096.0);
  }
  return t;
}

__constant__ Q1PointTab Q1_POINTS[8] = {q1_point_tab(0), q1_point_tab(1), q1_point_tab(2), q1_point_tab(3),
                                         q1_point_tab(4), q1_point_tab(5), q1_point_tab(6), q1_point_tab(7)};

// adj(J') and det' at the point described by tb (uniform)
__device__ __forceinline__ double q1_point_adj_rt(const double (&C)[2][2][2][3], const Q1PointTab& tb, double (&A)[3][3]) {
  const double xi = tb.xi[0], eta = tb.xi[1], zeta = tb.xi[2];
  double r0[3], r1[3], r2[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double m = fma(xi, C[1][1][1][c], C[1][1][0][c]);
    r0[c] = fma(zeta, fma(eta, C[1][1][1][c], C[1][0][1][c]), fma(eta, C[0][1][1][c], C[0][0][1][c]));
    r1[c] = fma(zeta, m, fma(xi, C[0][1][1][c], C[0][1][0][c]));
    r2[c] = fma(eta, m, fma(xi, C[1][0][1][c], C[1][0][0][c]));
  }
  A[0][0] = r1[1] * r2[2] - r1[2] * r2[1];
  A[0][1] = r1[2] * r2[0] - r1[0] * r2[2];
  A[0][2] = r1[0] * r2[1] - r1[1] * r2[0];
  A[1][0] = r2[1] * r0[2] - r2[2] * r0[1];
  A[1][1] = r2[2] * r0[0] - r2[0] * r0[2];
  A[1][2] = r2[0] * r0[1] - r2[1] * r0[0];
  A[2][0] = r0[1] * r1[2] - r0[2] * r1[1];
  A[2][1] = r0[2] * r1[0] - r0[0] * r1[2];
  A[2][2] = r0[0] * r1[1] - r0[1] * r1[0];
  return r0[0] * A[0][0] + r0[1] * A[0][1] + r0[2] * A[0][2];
}

// the 28 off-diagonal entries of the element Laplacian, rolled loop over the points
__device__ __forceinline__ void q1_laplace_lean_rolled(const double (&P)[2][2][2][3], const double ws, double (&L)[28]) {
  double C[2][2][2][3];
  q1_haar_coeffs(P, C);
#pragma unroll
  for (int i = 0; i < 28; ++i) L[i] = 0.0;
#pragma nounroll
  for (int G = 0; G < 8; ++G) {
    const Q1PointTab& tb = Q1_POINTS[G];
    double A[3][3];
    const double det = q1_point_adj_rt(C, tb, A);
    const double s = ws * q1_rcp(det);
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      double g[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) g[a] = fma(A[2][x], tb.h[2][a], fma(A[1][x], tb.h[1][a], A[0][x] * tb.h[0][a]));
#pragma unroll
      for (int a = 0; a < 7; ++a) {
        const double t = s * g[a];
#pragma unroll
        for (int b = a + 1; b < 8; ++b) L[q1_off(a, b)] = fma(t, g[b], L[q1_off(a, b)]);
      }
    }
  }
}

// ---- sum-factorised form of the same integral.  With M_g = (w / 512 det'_g) A A^T (symmetric 3 x 3, A = the cofactor rows of
// q1_point_adj) the element Laplacian is  L_ab = sum_g sum_de h_d[a](g) M_g,de h_e[b](g), and h_d[a](g) = s_d(a) prod_{f != d}
// (1 + s_f(a) xi_f(g)) is a tensor product of two-valued factors.  Per direction f the sum over its two Gauss abscissae is a 1-D
// transform of a pair (v-, v+):
//   quadratic factor (1 + s_f(a) xi_f)(1 + s_f(b) xi_f): three classes of the node pair -- both on the - side, both on the + side,
//   opposite sides:  [(1+g)^2 v- + (1-g)^2 v+,  (1-g)^2 v- + (1+g)^2 v+,  (2/3)(v- + v+)]  = (2/3) [2S - r3 D, 2S + r3 D, S]
//   linear factor 1 + s_f(a) xi_f: two classes (node on the - / + side):  [S - g D, S + g D],     S = v- + v+, D = v+ - v-.
// Diagonal terms d = e: the factor does not depend on xi_d (plain sum), quadratic in the two other directions: 8 -> 4 -> 6 -> 9 values.
// Mixed terms d < e (c the third direction): linear in e (node a) and d (node b), quadratic in c: 8 -> 8 -> 8 -> 12 values, and the
// (e, d) term is the (d, e) term with the nodes exchanged.  About 1,150 FP64 operations per element against 1,950 for the pointwise
// products of q1_laplace_lean (8 x (gradients 72 + scaled copies 21 + 84 pair products)).
constexpr double Q1_R3 = 1.73205080756887729353;   // sqrt(3)

__host__ __device__ constexpr int q1_bit(int a, int f) {   // side (0 / 1) of local node a in direction f
  constexpr int CX[8] = {0, 0, 1, 1, 0, 1, 1, 0};
  constexpr int CY[8] = {0, 1, 1, 0, 0, 0, 1, 1};
  constexpr int CZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  return f == 0 ? CX[a] : (f == 1 ? CY[a] : CZ[a]);
}
__host__ __device__ constexpr int q1_cls(int a, int b, int f) { return q1_bit(a, f) != q1_bit(b, f) ? 2 : q1_bit(a, f); }

__device__ __forceinline__ void q1_quad(const double vm, const double vp, double (&o)[3]) {   // x 3/2 (folded into the metric's scale)
  const double s = vm + vp, t = Q1_R3 * (vp - vm);
  o[0] = fma(2.0, s, -t);
  o[1] = fma(2.0, s, t);
  o[2] = s;
}
__device__ __forceinline__ void q1_lin(const double vm, const double vp, double (&o)[2]) {
  const double s = vm + vp, d = vp - vm;
  o[0] = fma(-Q1_GP, d, s);
  o[1] = fma(Q1_GP, d, s);
}

// scaled metric of Gauss point G (bit f of G = side in direction f): M[G] = {00, 11, 22, 01, 02, 12}
template <int G>
__device__ __forceinline__ void q1_point_metric(const double (&C)[2][2][2][3], const double ws, double (&M)[8][6]) {
  double A[3][3];
  const double det = q1_point_adj<(G & 1) * 2 - 1, ((G >> 1) & 1) * 2 - 1, ((G >> 2) & 1) * 2 - 1>(C, A);
  const double s = ws * q1_rcp(det);
  const double sd = s * (4.0 / 9.0), so = s * (2.0 / 3.0);   // two / one quadratic transforms follow
  M[G][0] = sd * fma(A[0][2], A[0][2], fma(A[0][1], A[0][1], A[0][0] * A[0][0]));
  M[G][1] = sd * fma(A[1][2], A[1][2], fma(A[1][1], A[1][1], A[1][0] * A[1][0]));
  M[G][2] = sd * fma(A[2][2], A[2][2], fma(A[2][1], A[2][1], A[2][0] * A[2][0]));
  M[G][3] = so * fma(A[0][2], A[1][2], fma(A[0][1], A[1][1], A[0][0] * A[1][0]));
  M[G][4] = so * fma(A[0][2], A[2][2], fma(A[0][1], A[2][1], A[0][0] * A[2][0]));
  M[G][5] = so * fma(A[1][2], A[2][2], fma(A[1][1], A[2][1], A[1][0] * A[2][0]));
}

template <int D, bool FIRST>
__device__ __forceinline__ void q1_sumfac_diag(const double (&M)[8][6], double (&L)[28]) {
  constexpr int F1 = D == 0 ? 1 : 0, F2 = D == 2 ? 1 : 2;
  double t[2][3], T[3][3];
#pragma unroll
  for (int b2 = 0; b2 < 2; ++b2) {
    const double v0 = M[(b2 << F2)][D] + M[(1 << D) | (b2 << F2)][D];
    const double v1 = M[(1 << F1) | (b2 << F2)][D] + M[(1 << D) | (1 << F1) | (b2 << F2)][D];
    q1_quad(v0, v1, t[b2]);
  }
#pragma unroll
  for (int g1 = 0; g1 < 3; ++g1) q1_quad(t[0][g1], t[1][g1], T[g1]);
#pragma unroll
  for (int a = 0; a < 7; ++a)
#pragma unroll
    for (int b = a + 1; b < 8; ++b) {
      const double v = T[q1_cls(a, b, F1)][q1_cls(a, b, F2)];
      const bool pos = q1_bit(a, D) == q1_bit(b, D);
      if (FIRST) L[q1_off(a, b)] = pos ? v : -v;
      else L[q1_off(a, b)] = pos ? L[q1_off(a, b)] + v : L[q1_off(a, b)] - v;
    }
}

template <int D, int E>
__device__ __forceinline__ void q1_sumfac_mixed(const double (&M)[8][6], double (&L)[28]) {
  constexpr int Cc = 3 - D - E, K = D == 0 ? 2 + E : 5;
  double u[2][2][2], w[2][2][2], U[2][2][3];
#pragma unroll
  for (int bd = 0; bd < 2; ++bd)
#pragma unroll
    for (int bc = 0; bc < 2; ++bc) q1_lin(M[(bd << D) | (bc << Cc)][K], M[(bd << D) | (1 << E) | (bc << Cc)][K], u[bd][bc]);   // [bd][bc][se]
#pragma unroll
  for (int bc = 0; bc < 2; ++bc)
#pragma unroll
    for (int se = 0; se < 2; ++se) q1_lin(u[0][bc][se], u[1][bc][se], w[bc][se]);   // [bc][se][sd]
#pragma unroll
  for (int se = 0; se < 2; ++se)
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) q1_quad(w[0][se][sd], w[1][se][sd], U[se][sd]);   // [se][sd][class in c]
#pragma unroll
  for (int a = 0; a < 7; ++a)
#pragma unroll
    for (int b = a + 1; b < 8; ++b) {
      const int gc = q1_cls(a, b, Cc);
      const double v1 = U[q1_bit(a, E)][q1_bit(b, D)][gc];   // s_d(a) s_e(b) h-factors: e on node a, d on node b
      const double v2 = U[q1_bit(b, E)][q1_bit(a, D)][gc];   // ... and the (e, d) term
      const bool p1 = q1_bit(a, D) == q1_bit(b, E), p2 = q1_bit(a, E) == q1_bit(b, D);
      double acc = L[q1_off(a, b)];
      acc = p1 ? acc + v1 : acc - v1;
      acc = p2 ? acc + v2 : acc - v2;
      L[q1_off(a, b)] = acc;
    }
}

// the 28 off-diagonal entries of the element Laplacian, sum-factorised
__device__ __forceinline__ void q1_laplace_sumfac(const double (&P)[2][2][2][3], const double ws, double (&L)[28]) {
  double C[2][2][2][3], M[8][6];
  q1_haar_coeffs(P, C);
  q1_point_metric<0>(C, ws, M);
  q1_point_metric<1>(C, ws, M);
  q1_point_metric<2>(C, ws, M);
  q1_point_metric<3>(C, ws, M);
  q1_point_metric<4>(C, ws, M);
  q1_point_metric<5>(C, ws, M);
  q1_point_metric<6>(C, ws, M);
  q1_point_metric<7>(C, ws, M);
  q1_sumfac_diag<0, true>(M, L);
  q1_sumfac_diag<1, false>(M, L);
  q1_sumfac_diag<2, false>(M, L);
  q1_sumfac_mixed<0, 1>(M, L);
  q1_sumfac_mixed<0, 2>(M, L);
  q1_sumfac_mixed<1, 2>(M, L);
}

template <int G>   // Gauss point G of the 2x2x2 rule: bit d of G = side of axis d
__device__ __forceinline__ void q1_laplace_point_idx(const double (&C)[2][2][2][3], const double ws, double (&L)[28]) {
  q1_laplace_point<(G & 1) * 2 - 1, ((G >> 1) & 1) * 2 - 1, ((G >> 2) & 1) * 2 - 1>(C, ws, L);
}

__device__ __forceinline__ double q1_sym(const double (&L)[28], int a, int b) { return a < b ? L[q1_off(a, b)] : L[q1_off(b, a)]; }

// the packed upper triangle L[36] (index tri(a, b)) of the element Laplacian from corner coordinates X[a][c] in the reference
// node order: lean closed form of the 2x2x2 rule, diagonal from the zero row sums
__device__ __forceinline__ void q1_laplace_lean36(const double (&X)[8][3], double (&L)[36]) {
  constexpr int CX[8] = {0, 0, 1, 1, 0, 1, 1, 0};
  constexpr int CY[8] = {0, 1, 1, 0, 0, 0, 1, 1};
  constexpr int CZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  double P[2][2][2][3], Lo[28];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) P[CZ[a]][CY[a]][CX[a]][c] = X[a][c];
  q1_laplace_sumfac(P, 1.0 / 512.0, Lo);
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    double d = 0.0;
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if (b != a) d -= q1_sym(Lo, a, b);
    L[tri(a, a)] = d;
#pragma unroll
    for (int b = a + 1; b < 8; ++b) L[tri(a, b)] = Lo[q1_off(a, b)];
  }
}

}  // namespace
