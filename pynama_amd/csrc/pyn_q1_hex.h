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

}  // namespace
