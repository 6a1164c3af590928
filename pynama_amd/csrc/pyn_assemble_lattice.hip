// Plan-free assembly kernels for Q1 hexahedral meshes with STRUCTURED topology (scalar Laplacian and the KLE blocks),
// the lattice detection of pyn_mesh_set and their launchers.  See DESIGN.md 4/5.
#include "pyn_internal.h"
#include "pyn_q1_hex.h"
#include "pyn_lattice.h"

namespace {


// Lean integration for meshes whose elements are ALL parallelepipeds (every box mesh the reference creates,
// src/domain/dmplex.py:8-21): four corner loads instead of eight, J = S.E from the three edge vectors
// (S[d][m] = sum_c hcoo[d][c] C_m[c], a table constant), L_ab = detJ sum_{r<=s} Q_rs T_rs[ab]; no quadrature
// loop, no affinity test: ~110 VGPRs instead of ~170, i.e. 4 instead of 2-3 waves per SIMD to hide the gather
// and store latencies.  lat_affine_L: the 36 upper-triangle entries of L_e for the element whose lowest corner is
// node n00 of plane gl (shared with the matrix-free operator below).
// inverse Jacobian and determinant of the parallelepiped element whose lowest corner is node n00 of plane gl
// J = S.E from the three edge vectors of a parallelepiped, its inverse and determinant
__device__ __forceinline__ double lat_affine_inv(const double* __restrict__ S, const double (&E)[3][3], double (&Ji)[3][3]) {
  double J[3][3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int x = 0; x < 3; ++x) J[d][x] = fma(S[d * 3 + 2], E[2][x], fma(S[d * 3 + 1], E[1][x], S[d * 3] * E[0][x]));
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

// the four corners that span a parallelepiped: origin and its x, y, z lattice neighbours (raw loads, no arithmetic: the caller may
// issue them long before it needs them)
__device__ __forceinline__ void lat_affine_corners(const LatArgs& T, int n00, int gl, double (&C4)[4][3]) {
  const int nx = T.nx;
  const double* q0 = T.xyz + (int64_t)(lat_plane(T, gl) + n00) * 3;
  const double* qz = T.xyz + (int64_t)(lat_plane(T, gl + 1) + n00) * 3;
  if (T.ablate == 32) {   // diagnostics: no coordinate loads at all (a fixed cube): what their latency costs the tile kernel
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int x = 0; x < 3; ++x) C4[d][x] = (d == x + 1) ? 0.005 : 0.0;
    return;
  }
#pragma unroll
  for (int x = 0; x < 3; ++x) {
    C4[0][x] = q0[x];
    C4[1][x] = q0[3 + x];
    C4[2][x] = q0[3 * nx + x];
    C4[3][x] = qz[x];
  }
}
__device__ __forceinline__ double lat_affine_geom_from(const double* __restrict__ S, const double (&C4)[4][3], double (&Ji)[3][3]) {
  double E[3][3];  // edge vectors along the lattice x, y, z directions
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int x = 0; x < 3; ++x) E[d][x] = C4[d + 1][x] - C4[0][x];
  return lat_affine_inv(S, E, Ji);
}
__device__ __forceinline__ double lat_affine_geom(const LatArgs& T, const double* __restrict__ S, int n00, int gl, double (&Ji)[3][3]) {
  double C4[4][3];
  lat_affine_corners(T, n00, gl, C4);
  return lat_affine_geom_from(S, C4, Ji);
}

__device__ __forceinline__ void lat_affine_L_from(const double (&Ji)[3][3], const double det, double (&L)[36]) {
  double Q[6];
  {
    constexpr int RS[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {0, 2}, {1, 2}};
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int a = RS[u][0], b = RS[u][1];
      Q[u] = det * (Ji[0][a] * Ji[0][b] + Ji[1][a] * Ji[1][b] + Ji[2][a] * Ji[2][b]);
    }
  }
  // 72 T_rs[ab] are small integers for the trilinear element (q1_aff_int): 15 products, then signed sums --
  // no table traffic at all inside the loop
  {
    double D[3][3], M[3][2];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const double qd = Q[u] * (1.0 / 72.0), qm = Q[3 + u] * (1.0 / 72.0);
      D[u][0] = 4.0 * qd;
      D[u][1] = 8.0 * qd;
      D[u][2] = 16.0 * qd;
      M[u][0] = 12.0 * qm;
      M[u][1] = 24.0 * qm;
    }
    int idx = 0;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int c = a; c < 8; ++c, ++idx) {
        double v = 0.0;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const int n = q1_aff_int(u, a, c);
          const int an = n < 0 ? -n : n;
          if (an == 0) continue;
          const double x = u < 3 ? D[u][an == 4 ? 0 : (an == 8 ? 1 : 2)] : M[u - 3][an == 12 ? 0 : 1];
          v = n > 0 ? v + x : v - x;
        }
        L[idx] = v;
      }
  }
}

__device__ __forceinline__ void lat_affine_L(const LatArgs& T, const double* __restrict__ S, int n00, int gl, double (&L)[36]) {
  double Ji[3][3];
  const double det = lat_affine_geom(T, S, n00, gl, Ji);
  lat_affine_L_from(Ji, det, L);
}

// does element t of the tile exist?  (its lattice position and the node id of its lowest corner)
template <int TX, int TY, int TZ>
__device__ __forceinline__ bool lat_tile_element(const LatArgs& T, int x0, int y0, int z0, int t, int& n00, int& gl) {
  using LT = LatTile<TX, TY, TZ>;
  const int lx = t % LT::EX, ly = (t / LT::EX) % LT::EY, lz = t / (LT::EX * LT::EY);
  const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
  gl = T.p_own0 + z0 - 1 + lz;
  n00 = gy * T.nx + gx;
  return t < LT::NE && gx >= 0 && gx < T.nx - 1 && gy >= 0 && gy < T.ny - 1 && gl >= 0 && gl < T.npl - 1;
}

// `pre`: the corners of the thread's FIRST element (t = t0) were requested by the caller before it cleared the accumulators (their
// latency then overlaps the clearing and its barrier); later elements of the thread (tiles with more than nt elements) load here
template <int TX, int TY, int TZ>
__device__ __forceinline__ void lat_integrate_affine(const LatArgs& T, int x0, int y0, int z0, double* acc, int t0, int nt,
                                                     const double (&pre)[4][3]) {
  using LT = LatTile<TX, TY, TZ>;
  const int nx = T.nx, ny = T.ny;
  const double* __restrict__ S = T.q.aff + 248;
  constexpr int CX[8] = {0, 0, 1, 1, 0, 1, 1, 0};
  constexpr int CY[8] = {0, 1, 1, 0, 0, 0, 1, 1};
  constexpr int CZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  for (int t = t0; t < LT::NE; t += nt) {
    const int lx = t % LT::EX, ly = (t / LT::EX) % LT::EY, lz = t / (LT::EX * LT::EY);
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly, gl = T.p_own0 + z0 - 1 + lz;
    if (gx < 0 || gx >= nx - 1 || gy < 0 || gy >= ny - 1 || gl < 0 || gl >= T.npl - 1) continue;
    const int n00 = gy * nx + gx;
    double L[36];
    if (t == t0) {
      double Ji[3][3];
      const double det = lat_affine_geom_from(S, pre, Ji);
      lat_affine_L_from(Ji, det, L);
    } else {
      lat_affine_L(T, S, n00, gl, L);
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int rx = lx - 1 + CX[a], ry = ly - 1 + CY[a], rz = lz - 1 + CZ[a];
      if (rx < 0 || rx >= TX || ry < 0 || ry >= TY || rz < 0 || rz >= TZ || z0 + rz >= T.n_own) continue;
      double* row = acc + ((rz * TY + ry) * TX + rx) * 27;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int kk = (CZ[c] - CZ[a] + 1) * 9 + (CY[c] - CY[a] + 1) * 3 + (CX[c] - CX[a] + 1);
        atomicAdd(&row[kk], L[tri(a, c)]);
      }
    }
  }
}

// one-off check behind std_lat: the closed-form row offsets equal the symbolic phase's rowptr
__global__ void lattice_rowptr_check_kernel(LatArgs T, const int32_t* __restrict__ rowptr, int64_t n_rows, int* flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const int x = (int)(i % T.nx), y = (int)((i / T.nx) % T.ny), zo = (int)(i / ((int64_t)T.nx * T.ny));
  if (rowptr[i] != lat_rowptr_std(T, x, y, zo)) *flag = 0;
}


// one tile per workgroup
template <int TX, int TY, int TZ, bool AFF>
__global__ void __launch_bounds__(TILE_THREADS, AFF ? 3 : 2) assemble_q1_hex_lattice_kernel(LatArgs T) {
  using LT = LatTile<TX, TY, TZ>;
  extern __shared__ __align__(16) double lds[];
  double* acc = lds;
  int* rlo = reinterpret_cast<int*>(acc + LT::ACC);
  int* zrd = rlo + LT::NR;
  unsigned char* nbc = reinterpret_cast<unsigned char*>(zrd + TZ);
  const int tid = threadIdx.x;
  const int b = T.ablate == 6 ? (int)blockIdx.x : xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int bx = b % T.ntx, by = (b / T.ntx) % T.nty, bz = b / (T.ntx * T.nty);
  const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
  // parallelepipeds: the corner coordinates of this thread's element are requested FIRST; their latency (the largest single item of
  // a tile's lifetime: with PYNAMA_LATTICE_ABLATE=32, no coordinate loads, the kernel runs 12-16 % shorter) then overlaps the
  // clearing of the accumulators (-2 % in a same-process A/B)
  double C4[4][3];
  if (AFF && T.ablate != 1) {
    int n00, gl;
    if (lat_tile_element<TX, TY, TZ>(T, x0, y0, z0, tid, n00, gl)) lat_affine_corners(T, n00, gl, C4);
  }
  LatMeta<TX, TY, TZ, TILE_THREADS> meta;
  lat_meta_load<TX, TY, TZ, TILE_THREADS>(T, x0, y0, z0, tid, meta);   // in flight during the element phase
  for (int i = tid; i < LT::ACC; i += TILE_THREADS) acc[i] = 0.0;
  __syncthreads();
  if (T.ablate != 1) {
    if (AFF)
      lat_integrate_affine<TX, TY, TZ>(T, x0, y0, z0, acc, tid, TILE_THREADS, C4);
    else
      lat_integrate<TX, TY, TZ>(T, x0, y0, z0, acc, tid, TILE_THREADS);
  }
  const int anybc = __syncthreads_or(lat_meta_commit<TX, TY, TZ, TILE_THREADS>(T, z0, tid, meta, rlo, zrd, nbc));
  lat_emit_dinv<TX, TY, TZ>(T, x0, y0, z0, acc, rlo, anybc ? nbc : nullptr, tid, TILE_THREADS);
  if (lat_tile_plain<TX, TY, TZ>(T, x0, y0, z0, zrd, anybc) && T.ablate != 4)
    lat_store_plain<TX, TY, TZ>(T, acc, rlo, tid, TILE_THREADS);
  else
    lat_store<TX, TY, TZ>(T, x0, y0, acc, rlo, zrd, nbc, tid, TILE_THREADS);
}

// y_e = L_e x_e for a parallelepiped WITHOUT forming L_e: in the Haar basis of each axis ((v0, v1) -> (s, d) = (v0 + v1,
// v1 - v0)) the 1-D factors of the trilinear element are monomial -- mass 2/3 [[1, 1/2], [1/2, 1]] -> diag(3, 1)/3... i.e.
// int N N -> (s, d) |-> (3 s, d)/3, int N' N' -> (0, d), int N' N -> s |-> d', its transpose d |-> s' -- so
//   L_e = H^-1 [ diagonal from Q_xx, Q_yy, Q_zz  +  12 symmetric couplings from Q_xy, Q_xz, Q_yz ] H,   Q = detJ J^-1 J^-T:
// 24 + 24 additions for H and H^-1, 7 products + 12 FMAs in between (~85 FP64 operations instead of ~280 for the 36
// entries + 64 for the product).  Corner a sits at (i, j, k) = (CX, CY, CZ)[a] (checked by aff_standard).
struct HaarQ {   // detJ J^-1 J^-T / 8 (the 1/8 of the three inverse transforms)
  double xx, yy, zz, xy, xz, yz;
};
__device__ __forceinline__ HaarQ haar_q(const double (&Ji)[3][3], const double det) {
  const double d8 = 0.125 * det;
  HaarQ q;
  q.xx = d8 * (Ji[0][0] * Ji[0][0] + Ji[1][0] * Ji[1][0] + Ji[2][0] * Ji[2][0]);
  q.yy = d8 * (Ji[0][1] * Ji[0][1] + Ji[1][1] * Ji[1][1] + Ji[2][1] * Ji[2][1]);
  q.zz = d8 * (Ji[0][2] * Ji[0][2] + Ji[1][2] * Ji[1][2] + Ji[2][2] * Ji[2][2]);
  q.xy = d8 * (Ji[0][0] * Ji[0][1] + Ji[1][0] * Ji[1][1] + Ji[2][0] * Ji[2][1]);
  q.xz = d8 * (Ji[0][0] * Ji[0][2] + Ji[1][0] * Ji[1][2] + Ji[2][0] * Ji[2][2]);
  q.yz = d8 * (Ji[0][1] * Ji[0][2] + Ji[1][1] * Ji[1][2] + Ji[2][1] * Ji[2][2]);
  return q;
}
constexpr int HAAR_NODE[2][2][2] = {{{0, 4}, {1, 7}}, {{3, 5}, {2, 6}}};   // [i][j][k] -> local node

// forward transform x, y, z: g[a][b][c], index 0 = s, 1 = d;  stride S between the 8 inputs (1: scalar, 3: component of a vector)
template <int S>
__device__ __forceinline__ void haar_fwd(const double* xe, double (&g)[2][2][2]) {
  double h[2][2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double v0 = xe[HAAR_NODE[0][j][k] * S], v1 = xe[HAAR_NODE[1][j][k] * S];
      g[0][j][k] = v0 + v1;
      g[1][j][k] = v1 - v0;
    }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double v0 = g[i][0][k], v1 = g[i][1][k];
      h[i][0][k] = v0 + v1;
      h[i][1][k] = v1 - v0;
    }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const double v0 = h[i][j][0], v1 = h[i][j][1];
      g[i][j][0] = v0 + v1;
      g[i][j][1] = v1 - v0;
    }
}

// the Laplacian in the Haar domain: 7 diagonal terms + 12 symmetric couplings
__device__ __forceinline__ void haar_laplace(const HaarQ& q, const double (&g)[2][2][2], double (&h)[2][2][2]) {
  const double third = 1.0 / 3.0;
  const double sxy = (q.xx + q.yy) * third, sxz = (q.xx + q.zz) * third, syz = (q.yy + q.zz) * third;
  const double sall = (q.xx + q.yy + q.zz) * (1.0 / 9.0);
  const double pxy = q.xy * third, pxz = q.xz * third, pyz = q.yz * third;
  h[0][0][0] = 0.0;
  h[1][0][0] = fma(q.xz, g[0][0][1], fma(q.xy, g[0][1][0], q.xx * g[1][0][0]));
  h[0][1][0] = fma(q.yz, g[0][0][1], fma(q.xy, g[1][0][0], q.yy * g[0][1][0]));
  h[0][0][1] = fma(q.yz, g[0][1][0], fma(q.xz, g[1][0][0], q.zz * g[0][0][1]));
  h[1][1][0] = fma(pyz, g[1][0][1], fma(pxz, g[0][1][1], sxy * g[1][1][0]));
  h[1][0][1] = fma(pyz, g[1][1][0], fma(pxy, g[0][1][1], sxz * g[1][0][1]));
  h[0][1][1] = fma(pxz, g[1][1][0], fma(pxy, g[1][0][1], syz * g[0][1][1]));
  h[1][1][1] = sall * g[1][1][1];
}

// inverse transform: o0 = s' - d', o1 = s' + d' per axis (the halves are in HaarQ); h is clobbered
template <int S>
__device__ __forceinline__ void haar_inv(double (&h)[2][2][2], double* ye) {
  double g[2][2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const double s0 = h[i][j][0], d0 = h[i][j][1];
      g[i][j][0] = s0 - d0;
      g[i][j][1] = s0 + d0;
    }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double s0 = g[i][0][k], d0 = g[i][1][k];
      h[i][0][k] = s0 - d0;
      h[i][1][k] = s0 + d0;
    }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double s0 = h[0][j][k], d0 = h[1][j][k];
      ye[HAAR_NODE[0][j][k] * S] = s0 - d0;
      ye[HAAR_NODE[1][j][k] * S] = s0 + d0;
    }
}

__device__ __forceinline__ void lat_affine_apply(const double (&Ji)[3][3], const double det, const double (&xe)[8], double (&ye)[8]) {
  const HaarQ q = haar_q(Ji, det);
  double g[2][2][2], h[2][2][2];
  haar_fwd<1>(xe, g);
  haar_laplace(q, g, h);
  haar_inv<1>(h, ye);
}

// KLE element product on a parallelepiped, everything in the Haar domain: the reduced (centroid) rule sees only the
// first-order coefficients -- the reference gradient of component q at the centroid is g_q[d along r] / 8 -- so
//   D = J^-1 gref (velocity gradient), W = c aw (D - D^T) + c ad tr(D) I, V = J^-T W, and h_p[d along r] += V[r][p] / 8
// adds the div/curl penalty terms to the Laplacian of component p before the inverse transform.
__device__ __forceinline__ void lat_affine_apply_kle(const double (&Ji)[3][3], const double det, const double cr, const double alpha_d,
                                                     const double alpha_w, const double (&xe)[8][3], double (&ye)[8][3]) {
  const HaarQ q = haar_q(Ji, det);
  double g[3][2][2][2];
#pragma unroll
  for (int p = 0; p < 3; ++p) haar_fwd<3>(&xe[0][p], g[p]);
  double gref[3][3];   // [r][q]
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    gref[0][p] = 0.125 * g[p][1][0][0];
    gref[1][p] = 0.125 * g[p][0][1][0];
    gref[2][p] = 0.125 * g[p][0][0][1];
  }
  double D[3][3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int p = 0; p < 3; ++p) D[d][p] = fma(Ji[d][2], gref[2][p], fma(Ji[d][1], gref[1][p], Ji[d][0] * gref[0][p]));
  const double caw = cr * alpha_w, tr = cr * alpha_d * (D[0][0] + D[1][1] + D[2][2]);
  double W[3][3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int p = 0; p < 3; ++p) W[d][p] = d == p ? tr : caw * (D[d][p] - D[p][d]);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    double h[2][2][2];
    haar_laplace(q, g[p], h);
    h[1][0][0] = fma(0.125, fma(Ji[2][0], W[2][p], fma(Ji[1][0], W[1][p], Ji[0][0] * W[0][p])), h[1][0][0]);
    h[0][1][0] = fma(0.125, fma(Ji[2][1], W[2][p], fma(Ji[1][1], W[1][p], Ji[0][1] * W[0][p])), h[0][1][0]);
    h[0][0][1] = fma(0.125, fma(Ji[2][2], W[2][p], fma(Ji[1][2], W[1][p], Ji[0][2] * W[0][p])), h[0][0][1]);
    haar_inv<3>(h, &ye[0][p]);
  }
}

// ---- matrix-free operator: y = A x for the scalar Laplacian WITHOUT the assembled matrix -------------------------
// A is the matrix pyn_assemble_scalar(LAPLACE) builds with the current Dirichlet mask (imposed rows = identity,
// imposed columns eliminated; base_problem.py:531-549 semantics): y_i = x_i on imposed rows, else
// y_i = sum_e sum_c L_e[a_i, c] x_c over free nodes c.  Same tile scheme as the assembly: a workgroup owns
// TX x TY x TZ rows, loads the (TX+2)(TY+2)(TZ+2) node box of x once into LDS (imposed nodes as 0), lets one lane
// per element form y_e = L_e x_e (parallelepipeds: lat_affine_apply, L_e never formed; general geometry: c G^T (G x_e) per Gauss point, no L_e at all) and adds the rows the
// tile owns with ds_add_f64; every y is written once, p.Ap partials fused.  HBM traffic per row: x 8 B + y 8 B +
// xyz 24 B + flag 1 B instead of the 27 x 8 B of matrix values the SELL kernel streams.
template <int TX, int TY, int TZ>
struct MfTile {
  static constexpr int NR = TX * TY * TZ, EX = TX + 1, EY = TY + 1, EZ = TZ + 1, NE = EX * EY * EZ;
  static constexpr int BX = TX + 2, BY = TY + 2, BZ = TZ + 2, NB = BX * BY * BZ;
  static constexpr size_t BYTES = (size_t)(NB + NR) * sizeof(double) + ((NB + 7) & ~7);
};

// One Gauss point of the matrix-free apply: ye += w detJ G^T (G xe)
__device__ __forceinline__ void gauss_point_apply(const TileArgs& T, const int G, const double (&X)[8][3], const double (&xe)[8],
                                                  double (&ye)[8]) {
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
  // reference-space gradient of the interpolated x, pushed to physical space, scaled, pulled back
  double gr[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int a = 0; a < 8; ++a) gr[d] = fma(hr[d * 8 + a], xe[a], gr[d]);
  const double cw = T.w[G] * det;
  double gp[3], gb[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) gp[d] = cw * fma(Ji[d][2], gr[2], fma(Ji[d][1], gr[1], Ji[d][0] * gr[0]));
#pragma unroll
  for (int m = 0; m < 3; ++m) gb[m] = fma(Ji[2][m], gp[2], fma(Ji[1][m], gp[1], Ji[0][m] * gp[0]));
#pragma unroll
  for (int a = 0; a < 8; ++a) ye[a] = fma(hr[16 + a], gb[2], fma(hr[8 + a], gb[1], fma(hr[a], gb[0], ye[a])));
}

template <int TX, int TY, int TZ, bool AFF, bool DOT>
__global__ void __launch_bounds__(256) lattice_matfree_laplace_kernel(LatArgs T, const double* __restrict__ xin, double* __restrict__ yout,
                                                                       const int* __restrict__ flag, double* __restrict__ part,
                                                                       int n_tiles) {
  using MT = MfTile<TX, TY, TZ>;
  extern __shared__ __align__(16) double lds[];
  __shared__ double smd[4];
  if (flag && flag[0]) return;
  double* xs = lds;                    // [NB] node box of x, imposed nodes as 0
  double* acc = xs + MT::NB;           // [NR]
  unsigned char* nbc = reinterpret_cast<unsigned char*>(acc + MT::NR);   // [NB] Dirichlet flags
  const int tid = threadIdx.x;
  const int nx = T.nx, ny = T.ny;
  const double* __restrict__ S = AFF ? T.q.aff + 248 : nullptr;
  constexpr int CX[8] = {0, 0, 1, 1, 0, 1, 1, 0};
  constexpr int CY[8] = {0, 1, 1, 0, 0, 0, 1, 1};
  constexpr int CZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  double dot = 0.0;
  for (int tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {   // gridDim.x is a multiple of 8 or >= n_tiles: XCD kept
    const int b = xcd_contiguous_tile(tb, n_tiles);
    const int bx = b % T.ntx, by = (b / T.ntx) % T.nty, bz = T.bz0 + (b / (T.ntx * T.nty)) * T.bzs;
    const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
    // ---- node box
    for (int i = tid; i < MT::NB; i += 256) {
      const int qx = i % MT::BX, qy = (i / MT::BX) % MT::BY, qz = i / (MT::BX * MT::BY);
      const int x = x0 - 1 + qx, y = y0 - 1 + qy, pl = T.p_own0 + z0 - 1 + qz;
      double v = 0.0;
      unsigned char f = 0;
      if (x >= 0 && x < nx && y >= 0 && y < ny && pl >= 0 && pl < T.npl) {
        const int64_t node = lat_plane(T, pl) + y * nx + x;
        f = T.bcmask ? T.bcmask[node] : 0;
        v = f ? 0.0 : xin[node];
      }
      xs[i] = v;
      nbc[i] = f;
    }
    for (int i = tid; i < MT::NR; i += 256) acc[i] = 0.0;
    __syncthreads();
    // ---- elements
    for (int t = tid; t < MT::NE; t += 256) {
      const int lx = t % MT::EX, ly = (t / MT::EX) % MT::EY, lz = t / (MT::EX * MT::EY);
      const int gx = x0 - 1 + lx, gy = y0 - 1 + ly, gl = T.p_own0 + z0 - 1 + lz;
      if (gx < 0 || gx >= nx - 1 || gy < 0 || gy >= ny - 1 || gl < 0 || gl >= T.npl - 1) continue;
      const int n00 = gy * nx + gx;
      double xe[8], ye[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) xe[a] = xs[((lz + CZ[a]) * MT::BY + ly + CY[a]) * MT::BX + lx + CX[a]];
      if (AFF) {
        double Ji[3][3];
        const double det = lat_affine_geom(T, S, n00, gl, Ji);
        lat_affine_apply(Ji, det, xe, ye);
      } else {
        const int pb = lat_plane(T, gl) + n00, pt = lat_plane(T, gl + 1) + n00;
        double X[8][3];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const double* q = T.xyz + (int64_t)((CZ[a] ? pt : pb) + CY[a] * nx + CX[a]) * 3;
          X[a][0] = q[0];
          X[a][1] = q[1];
          X[a][2] = q[2];
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) ye[a] = 0.0;
#pragma nounroll
        for (int g = 0; g < 8; ++g) gauss_point_apply(T.q, g, X, xe, ye);
      }
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int rx = lx - 1 + CX[a], ry = ly - 1 + CY[a], rz = lz - 1 + CZ[a];
        if (rx < 0 || rx >= TX || ry < 0 || ry >= TY || rz < 0 || rz >= TZ) continue;
        atomicAdd(&acc[(rz * TY + ry) * TX + rx], ye[a]);
      }
    }
    __syncthreads();
    // ---- rows of the tile: each y written once
    for (int s = tid; s < MT::NR; s += 256) {
      const int rx = s % TX, ry = (s / TX) % TY, rz = s / (TX * TY);
      const int x = x0 + rx, y = y0 + ry, zo = z0 + rz;
      if (x >= nx || y >= ny || zo >= T.n_own) continue;
      const int64_t node = lat_plane(T, T.p_own0 + zo) + y * nx + x;
      const int bi = ((rz + 1) * MT::BY + ry + 1) * MT::BX + rx + 1;
      const double xv = nbc[bi] ? xin[node] : xs[bi];
      const double yv = nbc[bi] ? xv : acc[s];
      yout[node] = yv;
      if (DOT) dot = fma(yv, xv, dot);
    }
    __syncthreads();
  }
  if (DOT) {
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if ((tid & 63) == 0) smd[tid >> 6] = dot;
    __syncthreads();
    if (tid == 0) part[blockIdx.x] = smd[0] + smd[1] + smd[2] + smd[3];
  }
}

// Column-marching variant for parallelepipeds: 15 x 15 x TZ rows per tile, so the 16 x 16 element columns of the tile are
// exactly the 256 threads; a thread walks its column upwards.  Per element: 4 LDS reads (the top face; the bottom face is
// the previous element's top), 4 ds_add_f64 (a plane of rows is complete for this column once the element above it has been
// added: the top contributions are carried in registers), and all x / y index work is loop-invariant.
template <int TZ, bool DOT>
__global__ void __launch_bounds__(256) lattice_matfree_laplace_march_kernel(LatArgs T, const double* __restrict__ xin,
                                                                             double* __restrict__ yout, const int* __restrict__ flag,
                                                                             double* __restrict__ part, int n_tiles) {
  using MT = MfTile<15, 15, TZ>;
  extern __shared__ __align__(16) double lds[];
  __shared__ double smd[4];
  if (flag && flag[0]) return;
  double* xs = lds;
  double* acc = xs + MT::NB;
  unsigned char* nbc = reinterpret_cast<unsigned char*>(acc + MT::NR);
  const int tid = threadIdx.x;
  const int nx = T.nx, ny = T.ny;
  const double* __restrict__ S = T.q.aff + 248;
  const int lx = tid & 15, ly = tid >> 4;
  // bottom-face corners in closure order: (0,0) (0,1) (1,1) (1,0); the top-face corner above bottom corner c is TOP[c]
  constexpr int BXo[4] = {0, 0, 1, 1}, BYo[4] = {0, 1, 1, 0}, TOP[4] = {4, 7, 6, 5};
  int boff[4], roff[4];
  bool rok[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    boff[c] = (ly + BYo[c]) * MT::BX + lx + BXo[c];
    const int rx = lx - 1 + BXo[c], ry = ly - 1 + BYo[c];
    rok[c] = rx >= 0 && rx < 15 && ry >= 0 && ry < 15;
    roff[c] = ry * 15 + rx;
  }
  double dot = 0.0;
  for (int tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {
    const int b = T.ablate == 6 ? tb : xcd_contiguous_tile(tb, n_tiles);
    const int bx = b % T.ntx, by = (b / T.ntx) % T.nty, bz = T.bz0 + (b / (T.ntx * T.nty)) * T.bzs;
    const int x0 = bx * 15, y0 = by * 15, z0 = bz * TZ;
    {   // node box: all loads of a thread issued before the first LDS write
      constexpr int NJ = (MT::NB + 255) / 256;
      double v[NJ];
      unsigned char f[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int i = tid + 256 * j;
        const int qx = i % MT::BX, qy = (i / MT::BX) % MT::BY, qz = i / (MT::BX * MT::BY);
        const int x = x0 - 1 + qx, y = y0 - 1 + qy, pl = T.p_own0 + z0 - 1 + qz;
        v[j] = 0.0;
        f[j] = 0;
        if (i < MT::NB && x >= 0 && x < nx && y >= 0 && y < ny && pl >= 0 && pl < T.npl) {
          const int64_t node = lat_plane(T, pl) + y * nx + x;
          f[j] = T.bcmask ? T.bcmask[node] : 0;
          v[j] = xin[node];
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int i = tid + 256 * j;
        if (i < MT::NB) {
          xs[i] = f[j] ? 0.0 : v[j];
          nbc[i] = f[j];
        }
      }
    }
    for (int i = tid; i < MT::NR; i += 256) acc[i] = 0.0;
    __syncthreads();
    {
      const int gx = x0 - 1 + lx, gy = y0 - 1 + ly;
      const bool col_ok = gx >= 0 && gx < nx - 1 && gy >= 0 && gy < ny - 1;
      const int n00 = gy * nx + gx;
      double xb[4], carry[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int c = 0; c < 4; ++c) xb[c] = xs[boff[c]];
#pragma nounroll
      for (int lz = 0; lz <= TZ; ++lz) {
        const int gl = T.p_own0 + z0 - 1 + lz;
        double xe[8], ye[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          xe[c] = xb[c];
          xe[TOP[c]] = xs[(lz + 1) * (MT::BX * MT::BY) + boff[c]];
        }
        if (col_ok && gl >= 0 && gl < T.npl - 1 && T.ablate != 1) {
          double Ji[3][3];
          const double det = lat_affine_geom(T, S, n00, gl, Ji);
          lat_affine_apply(Ji, det, xe, ye);
        } else {
#pragma unroll
          for (int a = 0; a < 8; ++a) ye[a] = 0.0;
        }
        if (lz >= 1) {   // row plane lz - 1 is complete for this column
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (rok[c]) atomicAdd(&acc[(lz - 1) * 225 + roff[c]], carry[c] + ye[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          carry[c] = ye[TOP[c]];
          xb[c] = xe[TOP[c]];
        }
      }
    }
    __syncthreads();
    for (int s = tid; s < MT::NR; s += 256) {
      const int rx = s % 15, ry = (s / 15) % 15, rz = s / 225;
      const int x = x0 + rx, y = y0 + ry, zo = z0 + rz;
      if (x >= nx || y >= ny || zo >= T.n_own) continue;
      const int64_t node = lat_plane(T, T.p_own0 + zo) + y * nx + x;
      const int bi = ((rz + 1) * MT::BY + ry + 1) * MT::BX + rx + 1;
      const double xv = nbc[bi] ? xin[node] : xs[bi];
      const double yv = nbc[bi] ? xv : acc[s];
      yout[node] = yv;
      if (DOT) dot = fma(yv, xv, dot);
    }
    __syncthreads();
  }
  if (DOT) {
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if ((tid & 63) == 0) smd[tid >> 6] = dot;
    __syncthreads();
    if (tid == 0) part[blockIdx.x] = smd[0] + smd[1] + smd[2] + smd[3];
  }
}

// ---- plan-free KLE assembly on lattices of parallelepipeds: the scalar lattice kernel's scheme (index
// arithmetic, stencil-ordered LDS rows, straight x-line copies for interior tiles) with the four-wave closed-form
// element blocks of assemble_q1_hex_kle_affine_kernel.  LDS row of a node = [p][27 stencil slots][q] = exactly the
// node's block-CSR row when it has all 27 neighbours.
struct KleLatArgs {
  LatArgs L;              // A = K or Rw, Arhs = Krhs (K only, may be null); bcmask per DOF (3 per node)
  double alpha_d, alpha_w;
  const double *wr, *hrsr, *Hr, *hcoor;   // reduced (centroid) rule
  const double* Lel = nullptr;            // general geometry, K: [28][ne] off-diagonal element Laplacians (kle_elem_laplace_kernel)
  int64_t ne = 0;                         // elements of this rank's lattice (nx-1)(ny-1)(npl-1)
};

// ---- matrix-free KLE operator: y = K x (3 DOFs per node) without the assembled matrix --------------------------------
// K is what pyn_assemble_kle builds (spectral.py:131,152-153 + base_problem.py:531-549): per element
//   K[(a,p),(b,q)] = d_pq (L_ab + c aw G_a.G_b) + c (ad G_pa G_qb - aw G_qa G_pb),   G = reduced-point gradients, c = w_r detJ
// so with the velocity gradient at the centroid D[d][q] = sum_b G_db x_bq the element product is
//   y_ap = sum_b L_ab x_bp + sum_d G_da W_dp,   W = c aw (D - D^T) + c ad tr(D) I
// (the Laplacian on every component + a rank-9 correction: ~350 FMAs instead of a 24 x 24 block).  Dirichlet DOFs:
// imposed columns enter as 0, imposed rows return x.  Same tile scheme as the scalar kernel, 3 doubles per node.
__device__ __forceinline__ double jac_inverse(const double* __restrict__ hc, const double (&X)[8][3], double (&Ji)[3][3]) {
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

// reduced-rule part of the element product: ye += sum_d G_da W_dp
__device__ __forceinline__ void kle_reduced_apply(const KleLatArgs& T, const double (&Ji)[3][3], double det, const double (&xe)[8][3],
                                                  double (&ye)[8][3]) {
  const double* __restrict__ hr = T.hrsr;
  const double cr = T.wr[0] * det;
  const double caw = cr * T.alpha_w, cad = cr * T.alpha_d;
  double Gr[3][8];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int a = 0; a < 8; ++a) Gr[d][a] = fma(Ji[d][2], hr[16 + a], fma(Ji[d][1], hr[8 + a], Ji[d][0] * hr[a]));
  double D[3][3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      double sacc = 0.0;
#pragma unroll
      for (int b = 0; b < 8; ++b) sacc = fma(Gr[d][b], xe[b][q], sacc);
      D[d][q] = sacc;
    }
  const double tr = cad * (D[0][0] + D[1][1] + D[2][2]);
  double W[3][3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) W[d][pp] = d == pp ? tr : caw * (D[d][pp] - D[pp][d]);
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int pp = 0; pp < 3; ++pp)
      ye[a][pp] = fma(Gr[2][a], W[2][pp], fma(Gr[1][a], W[1][pp], fma(Gr[0][a], W[0][pp], ye[a][pp])));
}

template <int TX, int TY, int TZ, bool AFF, bool DOT>
__global__ void __launch_bounds__(256) lattice_matfree_kle_kernel(KleLatArgs K, const double* __restrict__ xin, double* __restrict__ yout,
                                                                   const int* __restrict__ flag, double* __restrict__ part, int n_tiles) {
  using MT = MfTile<TX, TY, TZ>;
  extern __shared__ __align__(16) double lds[];
  __shared__ double smd[4];
  if (flag && flag[0]) return;
  const LatArgs& T = K.L;
  double* xs = lds;                        // [NB][3] node box of x, imposed DOFs as 0
  double* acc = xs + MT::NB * 3;           // [NR][3]
  unsigned char* nbc = reinterpret_cast<unsigned char*>(acc + MT::NR * 3);   // [NB] bit q: DOF q imposed
  const int tid = threadIdx.x;
  const int nx = T.nx, ny = T.ny;
  const double* __restrict__ S = AFF ? T.q.aff + 248 : nullptr;
  constexpr int CX[8] = {0, 0, 1, 1, 0, 1, 1, 0};
  constexpr int CY[8] = {0, 1, 1, 0, 0, 0, 1, 1};
  constexpr int CZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  double dot = 0.0;
  for (int tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {
    const int b = xcd_contiguous_tile(tb, n_tiles);
    const int bx = b % T.ntx, by = (b / T.ntx) % T.nty, bz = T.bz0 + (b / (T.ntx * T.nty)) * T.bzs;
    const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
    for (int i = tid; i < MT::NB; i += 256) {
      const int qx = i % MT::BX, qy = (i / MT::BX) % MT::BY, qz = i / (MT::BX * MT::BY);
      const int x = x0 - 1 + qx, y = y0 - 1 + qy, pl = T.p_own0 + z0 - 1 + qz;
      double v[3] = {0.0, 0.0, 0.0};
      unsigned char f = 0;
      if (x >= 0 && x < nx && y >= 0 && y < ny && pl >= 0 && pl < T.npl) {
        const int64_t node = lat_plane(T, pl) + y * nx + x;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int fq = T.bcmask ? (T.bcmask[node * 3 + q] ? 1 : 0) : 0;
          f |= fq << q;
          v[q] = fq ? 0.0 : xin[node * 3 + q];
        }
      }
      xs[i * 3] = v[0];
      xs[i * 3 + 1] = v[1];
      xs[i * 3 + 2] = v[2];
      nbc[i] = f;
    }
    for (int i = tid; i < MT::NR * 3; i += 256) acc[i] = 0.0;
    __syncthreads();
    for (int t = tid; t < MT::NE; t += 256) {
      const int lx = t % MT::EX, ly = (t / MT::EX) % MT::EY, lz = t / (MT::EX * MT::EY);
      const int gx = x0 - 1 + lx, gy = y0 - 1 + ly, gl = T.p_own0 + z0 - 1 + lz;
      if (gx < 0 || gx >= nx - 1 || gy < 0 || gy >= ny - 1 || gl < 0 || gl >= T.npl - 1) continue;
      const int n00 = gy * nx + gx;
      double xe[8][3], ye[8][3];
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const double* q = xs + (((lz + CZ[a]) * MT::BY + ly + CY[a]) * MT::BX + lx + CX[a]) * 3;
        xe[a][0] = q[0];
        xe[a][1] = q[1];
        xe[a][2] = q[2];
      }
      if (AFF) {
        double Ji[3][3];
        const double det = lat_affine_geom(T, S, n00, gl, Ji);
        lat_affine_apply_kle(Ji, det, K.wr[0] * det, K.alpha_d, K.alpha_w, xe, ye);
      } else {
        const int pb = lat_plane(T, gl) + n00, pt = lat_plane(T, gl + 1) + n00;
        double X[8][3];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const double* q = T.xyz + (int64_t)((CZ[a] ? pt : pb) + CY[a] * nx + CX[a]) * 3;
          X[a][0] = q[0];
          X[a][1] = q[1];
          X[a][2] = q[2];
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) ye[a][0] = ye[a][1] = ye[a][2] = 0.0;
#pragma nounroll
        for (int g = 0; g < 8; ++g) {   // full rule: the Laplacian on every component, ye += w detJ G^T (G xe)
          const double* __restrict__ hr = T.q.hrs + g * 24;
          double Ji[3][3];
          const double cw = T.q.w[g] * jac_inverse(T.q.hcoo + g * 24, X, Ji);
#pragma unroll
          for (int pp = 0; pp < 3; ++pp) {
            double gr[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
              for (int a = 0; a < 8; ++a) gr[d] = fma(hr[d * 8 + a], xe[a][pp], gr[d]);
            double gp[3], gb[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) gp[d] = cw * fma(Ji[d][2], gr[2], fma(Ji[d][1], gr[1], Ji[d][0] * gr[0]));
#pragma unroll
            for (int m = 0; m < 3; ++m) gb[m] = fma(Ji[2][m], gp[2], fma(Ji[1][m], gp[1], Ji[0][m] * gp[0]));
#pragma unroll
            for (int a = 0; a < 8; ++a) ye[a][pp] = fma(hr[16 + a], gb[2], fma(hr[8 + a], gb[1], fma(hr[a], gb[0], ye[a][pp])));
          }
        }
        double Ji[3][3];
        const double det = jac_inverse(K.hcoor, X, Ji);
        kle_reduced_apply(K, Ji, det, xe, ye);
      }
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int rx = lx - 1 + CX[a], ry = ly - 1 + CY[a], rz = lz - 1 + CZ[a];
        if (rx < 0 || rx >= TX || ry < 0 || ry >= TY || rz < 0 || rz >= TZ) continue;
        double* row = acc + ((rz * TY + ry) * TX + rx) * 3;
        atomicAdd(&row[0], ye[a][0]);
        atomicAdd(&row[1], ye[a][1]);
        atomicAdd(&row[2], ye[a][2]);
      }
    }
    __syncthreads();
    for (int s = tid; s < MT::NR * 3; s += 256) {
      const int r = s / 3, q = s - r * 3;
      const int rx = r % TX, ry = (r / TX) % TY, rz = r / (TX * TY);
      const int x = x0 + rx, y = y0 + ry, zo = z0 + rz;
      if (x >= nx || y >= ny || zo >= T.n_own) continue;
      const int64_t node = lat_plane(T, T.p_own0 + zo) + y * nx + x;
      const int bi = ((rz + 1) * MT::BY + ry + 1) * MT::BX + rx + 1;
      const bool imp = (nbc[bi] >> q) & 1;
      const double xv = imp ? xin[node * 3 + q] : xs[bi * 3 + q];
      const double yv = imp ? xv : acc[s];
      yout[node * 3 + q] = yv;
      if (DOT) dot = fma(yv, xv, dot);
    }
    __syncthreads();
  }
  if (DOT) {
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if ((tid & 63) == 0) smd[tid >> 6] = dot;
    __syncthreads();
    if (tid == 0) part[blockIdx.x] = smd[0] + smd[1] + smd[2] + smd[3];
  }
}


template <int TX, int TY, int TZ, bool RW, int A0>
__device__ __forceinline__ void kle_lat_rows(const KleLatArgs& T, const double (&Ji)[3][3], double det, int lx, int ly, int lz,
                                             int z0, double* acc) {
  constexpr int CX[8] = {0, 0, 1, 1, 0, 1, 1, 0};
  constexpr int CY[8] = {0, 1, 1, 0, 0, 0, 1, 1};
  constexpr int CZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  const double cr = T.wr[0] * det;
  const double caw = cr * T.alpha_w, cad = cr * T.alpha_d;
  const double* __restrict__ hr = T.hrsr;
  double Gr[3][8];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int a = 0; a < 8; ++a) Gr[d][a] = fma(Ji[d][2], hr[16 + a], fma(Ji[d][1], hr[8 + a], Ji[d][0] * hr[a]));
  double D[3][3], M2[3][2], DM[3][3][3];
  if (!RW) {
    constexpr int RS[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {0, 2}, {1, 2}};
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int a0 = RS[u][0], a1 = RS[3 + u][0], b1 = RS[3 + u][1];
      const double qd = det * (Ji[0][a0] * Ji[0][a0] + Ji[1][a0] * Ji[1][a0] + Ji[2][a0] * Ji[2][a0]) * (1.0 / 72.0);
      const double qm = det * (Ji[0][a1] * Ji[0][b1] + Ji[1][a1] * Ji[1][b1] + Ji[2][a1] * Ji[2][b1]) * (1.0 / 72.0);
      D[u][0] = 4.0 * qd, D[u][1] = 8.0 * qd, D[u][2] = 16.0 * qd;
      M2[u][0] = 12.0 * qm, M2[u][1] = 24.0 * qm;
    }
  } else {
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double x = det * Ji[m][d] * (1.0 / 72.0);
        DM[m][d][0] = 4.0 * x, DM[m][d][1] = 8.0 * x, DM[m][d][2] = 16.0 * x;
      }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int a = A0 + h;
    const int rx = lx - 1 + CX[a], ry = ly - 1 + CY[a], rz = lz - 1 + CZ[a];
    if (rx < 0 || rx >= TX || ry < 0 || ry >= TY || rz < 0 || rz >= TZ || z0 + rz >= T.L.n_own) continue;
    double* rowp = acc + ((rz * TY + ry) * TX + rx) * 243;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int kk = (CZ[b] - CZ[a] + 1) * 9 + (CY[b] - CY[a] + 1) * 3 + (CX[b] - CX[a] + 1);
      if (!RW) {
        double lab = 0.0;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const int n = q1_aff_int(u, a, b);
          const int an = n < 0 ? -n : n;
          if (an == 0) continue;
          const double x = u < 3 ? D[u][an == 4 ? 0 : (an == 8 ? 1 : 2)] : M2[u - 3][an == 12 ? 0 : 1];
          lab = n > 0 ? lab + x : lab - x;
        }
        const double s_ab = Gr[0][a] * Gr[0][b] + Gr[1][a] * Gr[1][b] + Gr[2][a] * Gr[2][b];
        const double diag = lab + caw * s_ab;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            double v = cad * Gr[pp][a] * Gr[q][b] - caw * Gr[q][a] * Gr[pp][b];
            if (pp == q) v += diag;
            atomicAdd(&rowp[(pp * 27 + kk) * 3 + q], v);
          }
      } else {
        const double hb = T.Hr[b];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          double tv = 0.0;
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const int n = q1_mix_int(d, a, b);
            const int an = n < 0 ? -n : n;
            const double x = DM[m][d][an == 4 ? 0 : (an == 8 ? 1 : 2)];
            tv = n > 0 ? tv + x : tv - x;
          }
          const double wv = tv - caw * Gr[m][a] * hb;
          const int P1 = (m + 1) % 3, P2 = (m + 2) % 3;
          atomicAdd(&rowp[(P2 * 27 + kk) * 3 + P1], wv);
          atomicAdd(&rowp[(P1 * 27 + kk) * 3 + P2], -wv);
        }
      }
    }
  }
}

// General geometry (any trilinear hexahedron): the same block formulas with the element quantities from the closed forms of the
// 2x2x2 rule (pyn_q1_hex.h) instead of the parallelepiped integrals --
//   L_ab    = sum_g (w_g / 512) / det'_g  g_g[a] . g_g[b],                g = adj(J') h  (unnormalised gradients, G = g / det')
//   T_m[ab] = sum_g w_g detJ_g H_a(g) G_mb(g) = (1 / 4096) sum_g N'_a(g) g_g,m[b]      (the determinant cancels)
//   Gr      = adj(J'_0) s / det'_0,  c_r = w_r det'_0 / 512                            (reduced rule: the centroid)
// K (RW = false): L_ab comes from the per-element pre-pass below; the wave adds its two node ROWS {A0, A0+1}.
// Rw (RW = true), one Gauss point: this wave's two node COLUMNS {A0, A0+1}: Tc[m][a][h] += N'_a g_m[b_h] -- only the gradients of its
// two columns are formed (18 instead of 72 FP64 instructions per point); the determinant cancels and is never computed.
template <int A0>
__device__ __forceinline__ void kle_gen_point_rw(const double (&C)[2][2][2][3], const Q1PointTab& tb, double (&Tc)[3][8][2]) {
  double A[3][3];
  (void)q1_point_adj_rt(C, tb, A);
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = A0 + h;
      const double gb = fma(A[2][m], tb.h[2][b], fma(A[1][m], tb.h[1][b], A[0][m] * tb.h[0][b]));
#pragma unroll
      for (int a = 0; a < 8; ++a) Tc[m][a][h] = fma(tb.n[a], gb, Tc[m][a][h]);   // tb.n = N'_a / 4096 at this point
    }
}

// K: the scalar Laplacian part L_ab of an element does not depend on the tile that adds it.  It is integrated ONCE per element by
// a pre-pass (one element per lane, sum-factorised 2x2x2 rule, 28 off-diagonal entries, structure-of-arrays [28][ne]: 224 B per
// element) and read by the (up to eight) tiles whose rows the element touches -- the tile kernel then is the parallelepiped
// kernel plus 14 loads per wave, without the LDS exchange, its barriers and the 2.4-fold redundant integration of the first version.
__global__ void __launch_bounds__(256) kle_elem_laplace_kernel(LatArgs L, double* __restrict__ Lel, int64_t ne) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= ne) return;
  const int ex = L.nx - 1, ey = L.ny - 1;
  const int ix = (int)(e % ex), iy = (int)((e / ex) % ey), gl = (int)(e / ((int64_t)ex * ey));
  const int n00 = iy * L.nx + ix;
  const double* q0 = L.xyz + (int64_t)(lat_plane(L, gl) + n00) * 3;
  const double* qz = L.xyz + (int64_t)(lat_plane(L, gl + 1) + n00) * 3;
  double P[2][2][2][3], Lo[28];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        P[0][j][i][cc] = q0[(j * L.nx + i) * 3 + cc];
        P[1][j][i][cc] = qz[(j * L.nx + i) * 3 + cc];
      }
  q1_laplace_sumfac(P, 1.0 / 512.0, Lo);
#pragma unroll
  for (int i = 0; i < 28; ++i) Lel[(int64_t)i * ne + e] = Lo[i];
}

#ifndef KLE_RW_M_OUTER
#define KLE_RW_M_OUTER 1
#endif
// Rw with the vorticity component m as the OUTER loop: 16 instead of 48 accumulators live across the Gauss loop (a wave's two columns x
// eight rows for one m), the Jacobian rows are evaluated three times (8 x 3 x (24 + 6 + 6 + 16) = 1,250 instead of 860 FP64
// instructions per wave) -- but the kernel fits 168 registers, i.e. three workgroups per CU like the K kernel instead of two.
template <int TX, int TY, int TZ, int A0>
__device__ __forceinline__ void kle_lat_rw_general_m(const KleLatArgs& T, const double (&C)[2][2][2][3], int lx, int ly, int lz, int z0,
                                                     double* acc) {
  constexpr int CX[8] = {0, 0, 1, 1, 0, 1, 1, 0};
  constexpr int CY[8] = {0, 1, 1, 0, 0, 0, 1, 1};
  constexpr int CZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  // centroid: cofactor rows of J'_0 (first-order Haar coefficients); Gr[m][a] = (sum_d s_d(a) A0[d][m]) / det'_0
  double A0c[3][3], ri, caw;
  {
    double r0[3], r1[3], r2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      r0[c] = C[0][0][1][c];
      r1[c] = C[0][1][0][c];
      r2[c] = C[1][0][0][c];
    }
    A0c[0][0] = r1[1] * r2[2] - r1[2] * r2[1];
    A0c[0][1] = r1[2] * r2[0] - r1[0] * r2[2];
    A0c[0][2] = r1[0] * r2[1] - r1[1] * r2[0];
    A0c[1][0] = r2[1] * r0[2] - r2[2] * r0[1];
    A0c[1][1] = r2[2] * r0[0] - r2[0] * r0[2];
    A0c[1][2] = r2[0] * r0[1] - r2[1] * r0[0];
    A0c[2][0] = r0[1] * r1[2] - r0[2] * r1[1];
    A0c[2][1] = r0[2] * r1[0] - r0[0] * r1[2];
    A0c[2][2] = r0[0] * r1[1] - r0[1] * r1[0];
    const double det0 = r0[0] * A0c[0][0] + r0[1] * A0c[0][1] + r0[2] * A0c[0][2];
    ri = q1_rcp(det0);
    caw = T.wr[0] * det0 * (1.0 / 512.0) * T.alpha_w;
  }
  // LDS offsets of the element's eight node rows (-1: the row is not this tile's)
  int rofs[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const int rx = lx - 1 + CX[a], ry = ly - 1 + CY[a], rz = lz - 1 + CZ[a];
    rofs[a] = (rx < 0 || rx >= TX || ry < 0 || ry >= TY || rz < 0 || rz >= TZ || z0 + rz >= T.L.n_own) ? -1 : ((rz * TY + ry) * TX + rx) * 243;
  }
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    double Tm[8][2];
#pragma unroll
    for (int a = 0; a < 8; ++a) Tm[a][0] = Tm[a][1] = 0.0;
#pragma nounroll
    for (int G = 0; G < 8; ++G) {
      const Q1PointTab& tb = Q1_POINTS[G];
      const double xi = tb.xi[0], eta = tb.xi[1], zeta = tb.xi[2];
      double r0[3], r1[3], r2[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double mm = fma(xi, C[1][1][1][c], C[1][1][0][c]);
        r0[c] = fma(zeta, fma(eta, C[1][1][1][c], C[1][0][1][c]), fma(eta, C[0][1][1][c], C[0][0][1][c]));
        r1[c] = fma(zeta, mm, fma(xi, C[0][1][1][c], C[0][1][0][c]));
        r2[c] = fma(eta, mm, fma(xi, C[1][0][1][c], C[1][0][0][c]));
      }
      const int m1 = (m + 1) % 3, m2 = (m + 2) % 3;       // component m of the cross products r1 x r2, r2 x r0, r0 x r1
      const double a0 = r1[m1] * r2[m2] - r1[m2] * r2[m1];
      const double a1 = r2[m1] * r0[m2] - r2[m2] * r0[m1];
      const double a2 = r0[m1] * r1[m2] - r0[m2] * r1[m1];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int b = A0 + h;
        const double gb = fma(a2, tb.h[2][b], fma(a1, tb.h[1][b], a0 * tb.h[0][b]));
#pragma unroll
        for (int a = 0; a < 8; ++a) Tm[a][h] = fma(tb.n[a], gb, Tm[a][h]);
      }
    }
    const int P1 = (m + 1) % 3, P2 = (m + 2) % 3;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      if (rofs[a] < 0) continue;
      double* rowp = acc + rofs[a];
      const double gr = ((2 * CX[a] - 1) * A0c[0][m] + (2 * CY[a] - 1) * A0c[1][m] + (2 * CZ[a] - 1) * A0c[2][m]) * ri;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int b = A0 + h;
        const int kk = (CZ[b] - CZ[a] + 1) * 9 + (CY[b] - CY[a] + 1) * 3 + (CX[b] - CX[a] + 1);
        const double wv = Tm[a][h] - caw * gr * T.Hr[b];
        atomicAdd(&rowp[(P2 * 27 + kk) * 3 + P1], wv);
        atomicAdd(&rowp[(P1 * 27 + kk) * 3 + P2], -wv);
      }
    }
  }
}

template <int TX, int TY, int TZ, bool RW, int A0>
__device__ __forceinline__ void kle_lat_rows_general(const KleLatArgs& T, const double (&C)[2][2][2][3], const double (&Lv)[2][8], int lx,
                                                     int ly, int lz, int z0, double* acc) {
  constexpr int CX[8] = {0, 0, 1, 1, 0, 1, 1, 0};
  constexpr int CY[8] = {0, 1, 1, 0, 0, 0, 1, 1};
  constexpr int CZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  // reduced rule: the centroid (xi = 0): J'_0 rows are the first-order Haar coefficients
  double Gr[3][8], cr;
  {
    double r0[3], r1[3], r2[3], A[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      r0[c] = C[0][0][1][c];
      r1[c] = C[0][1][0][c];
      r2[c] = C[1][0][0][c];
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
    const double det0 = r0[0] * A[0][0] + r0[1] * A[0][1] + r0[2] * A[0][2];
    const double ri = q1_rcp(det0);
    cr = T.wr[0] * det0 * (1.0 / 512.0);
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int a = 0; a < 8; ++a)
        Gr[x][a] = ((2 * CX[a] - 1) * A[0][x] + (2 * CY[a] - 1) * A[1][x] + (2 * CZ[a] - 1) * A[2][x]) * ri;
  }
  const double caw = cr * T.alpha_w, cad = cr * T.alpha_d;
  double Lab[2][8], Tc[3][8][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      Lab[h][b] = 0.0;
      Tc[0][b][h] = Tc[1][b][h] = Tc[2][b][h] = 0.0;
    }
  if (RW) {
#pragma nounroll
    for (int G = 0; G < 8; ++G) kle_gen_point_rw<A0>(C, Q1_POINTS[G], Tc);
  } else {     // this wave's two rows of the shared element Laplacian; the diagonal from the zero row sums
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int a = A0 + h;
      double d = 0.0;
#pragma unroll
      for (int b = 0; b < 8; ++b)
        if (b != a) {
          const double v = Lv[h][b];
          Lab[h][b] = v;
          d -= v;
        }
      Lab[h][a] = d;
    }
  }
  if (!RW) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int a = A0 + h;
      const int rx = lx - 1 + CX[a], ry = ly - 1 + CY[a], rz = lz - 1 + CZ[a];
      if (rx < 0 || rx >= TX || ry < 0 || ry >= TY || rz < 0 || rz >= TZ || z0 + rz >= T.L.n_own) continue;
      double* rowp = acc + ((rz * TY + ry) * TX + rx) * 243;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int kk = (CZ[b] - CZ[a] + 1) * 9 + (CY[b] - CY[a] + 1) * 3 + (CX[b] - CX[a] + 1);
        const double s_ab = Gr[0][a] * Gr[0][b] + Gr[1][a] * Gr[1][b] + Gr[2][a] * Gr[2][b];
        const double diag = Lab[h][b] + caw * s_ab;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            double v = cad * Gr[pp][a] * Gr[q][b] - caw * Gr[q][a] * Gr[pp][b];
            if (pp == q) v += diag;
            atomicAdd(&rowp[(pp * 27 + kk) * 3 + q], v);
          }
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < 8; ++a) {     // every node row of the element, this wave's two columns
      const int rx = lx - 1 + CX[a], ry = ly - 1 + CY[a], rz = lz - 1 + CZ[a];
      if (rx < 0 || rx >= TX || ry < 0 || ry >= TY || rz < 0 || rz >= TZ || z0 + rz >= T.L.n_own) continue;
      double* rowp = acc + ((rz * TY + ry) * TX + rx) * 243;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int b = A0 + h;
        const int kk = (CZ[b] - CZ[a] + 1) * 9 + (CY[b] - CY[a] + 1) * 3 + (CX[b] - CX[a] + 1);
        const double hb = T.Hr[b];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const double wv = Tc[m][a][h] - caw * Gr[m][a] * hb;
          const int P1 = (m + 1) % 3, P2 = (m + 2) % 3;
          atomicAdd(&rowp[(P2 * 27 + kk) * 3 + P1], wv);
          atomicAdd(&rowp[(P1 * 27 + kk) * 3 + P2], -wv);
        }
      }
    }
  }
}

// loads of one element of the general-geometry KLE kernels: the eight corners and, for K, this wave's two rows {2 part, 2 part + 1} of
// the element Laplacian integrated by kle_elem_laplace_kernel (element id = x + (nx-1)(y + (ny-1) layer)); raw loads, no arithmetic
template <bool RW, int WHAT = 3>   // WHAT: 1 the Laplacian rows, 2 the corners, 3 both
__device__ __forceinline__ void kle_gen_loads(const KleLatArgs& T, int part, int n00, int gx, int gy, int gl, double (&P)[2][2][2][3],
                                              double (&Lv)[2][8]) {
  const LatArgs& L = T.L;
  const int nx = L.nx, ny = L.ny;
  const double* q0 = L.xyz + (int64_t)(lat_plane(L, gl) + n00) * 3;
  const double* qz = L.xyz + (int64_t)(lat_plane(L, gl + 1) + n00) * 3;
  if (!RW && (WHAT & 1)) {
    const double* Le = T.Lel + ((int64_t)gl * (ny - 1) + gy) * (nx - 1) + gx;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int bb = 0; bb < 8; ++bb) {
        const int a = 2 * part + h;
        const int lo = min(a, bb), hi = max(a, bb);
        const int off = lo * 7 - (lo * (lo - 1)) / 2 + (hi - lo - 1);     // q1_off(lo, hi)
        Lv[h][bb] = a == bb ? 0.0 : Le[(int64_t)off * T.ne];
      }
  }
  if (WHAT & 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          P[0][j][i][cc] = q0[(j * nx + i) * 3 + cc];
          P[1][j][i][cc] = qz[(j * nx + i) * 3 + cc];
        }
  }
}

// wave `part` adds the node rows (K) / node columns (Rw) {2 part, 2 part + 1} of its element
template <int TX, int TY, int TZ, bool RW>
__device__ __forceinline__ void kle_gen_dispatch(const KleLatArgs& T, int part, const double (&C)[2][2][2][3], const double (&Lv)[2][8], int lx,
                                                 int ly, int lz, int z0, double* acc) {
  if (RW && KLE_RW_M_OUTER) {
    switch (part) {
      case 0: kle_lat_rw_general_m<TX, TY, TZ, 0>(T, C, lx, ly, lz, z0, acc); break;
      case 1: kle_lat_rw_general_m<TX, TY, TZ, 2>(T, C, lx, ly, lz, z0, acc); break;
      case 2: kle_lat_rw_general_m<TX, TY, TZ, 4>(T, C, lx, ly, lz, z0, acc); break;
      default: kle_lat_rw_general_m<TX, TY, TZ, 6>(T, C, lx, ly, lz, z0, acc); break;
    }
    return;
  }
  switch (part) {
    case 0: kle_lat_rows_general<TX, TY, TZ, RW, 0>(T, C, Lv, lx, ly, lz, z0, acc); break;
    case 1: kle_lat_rows_general<TX, TY, TZ, RW, 2>(T, C, Lv, lx, ly, lz, z0, acc); break;
    case 2: kle_lat_rows_general<TX, TY, TZ, RW, 4>(T, C, Lv, lx, ly, lz, z0, acc); break;
    default: kle_lat_rows_general<TX, TY, TZ, RW, 6>(T, C, Lv, lx, ly, lz, z0, acc); break;
  }
}

template <int TX, int TY, int TZ, bool RW, bool GEN>
__global__ void __launch_bounds__(256, GEN && RW && !KLE_RW_M_OUTER ? 2 : 3) assemble_q1_hex_kle_lattice_kernel(KleLatArgs T) {
  using LT = LatTile<TX, TY, TZ>;
  constexpr int ROW = 243, ACC = LT::NR * ROW;
  extern __shared__ __align__(16) double lds[];
  double* acc = lds;
  int* rlo = reinterpret_cast<int*>(acc + ACC);
  int* zrd = rlo + LT::NR;
  unsigned char* nbc = reinterpret_cast<unsigned char*>(zrd + TZ);
  const LatArgs& L = T.L;
  const int tid = threadIdx.x, lane = tid & 63, part = tid >> 6;
  const int b = L.ablate == 6 ? (int)blockIdx.x : xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int bx = b % L.ntx, by = (b / L.ntx) % L.nty, bz = b / (L.ntx * L.nty);
  const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
  const int nx = L.nx, ny = L.ny;
  LatMeta<TX, TY, TZ, 256> meta;
  lat_meta_load<TX, TY, TZ, 256, 3>(L, x0, y0, z0, tid, meta);
  // the loads of a lane's FIRST element (for 3 x 3 x 3 tiles: its only one) are requested before the 52 KB of accumulators are cleared:
  // their latency overlaps the clearing and its barrier instead of following it
  // (K with general geometry keeps its own loop below: routed through the shared helpers its code lands on 168 VGPRs + spills and
  // runs 14 % longer; written out in place it needs 108)
  constexpr bool EARLY = !(GEN && !RW);
  double P[2][2][2][3], Lv[2][8], C4[4][3];
  bool pre_ok = false;
  if (!EARLY && L.ablate != 1) {   // K, general geometry: this wave's 14 Laplacian entries and the eight corners
    const int t = lane;
    const int lx = t % LT::EX, ly = (t / LT::EX) % LT::EY, lz = t / (LT::EX * LT::EY);
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly, gl = L.p_own0 + z0 - 1 + lz;
    if (t < LT::NE && gx >= 0 && gx < nx - 1 && gy >= 0 && gy < ny - 1 && gl >= 0 && gl < L.npl - 1)
      kle_gen_loads<RW, 3>(T, part, gy * nx + gx, gx, gy, gl, P, Lv);
  }
  if (EARLY && L.ablate != 1) {
    const int t = lane;
    const int lx = t % LT::EX, ly = (t / LT::EX) % LT::EY, lz = t / (LT::EX * LT::EY);
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly, gl = L.p_own0 + z0 - 1 + lz;
    pre_ok = t < LT::NE && gx >= 0 && gx < nx - 1 && gy >= 0 && gy < ny - 1 && gl >= 0 && gl < L.npl - 1;
    if (pre_ok) {
      const int n00 = gy * nx + gx;
      if (GEN) {
        kle_gen_loads<RW>(T, part, n00, gx, gy, gl, P, Lv);
      } else {
        lat_affine_corners(L, n00, gl, C4);
      }
    }
  }
  for (int i = tid; i < ACC; i += 256) acc[i] = 0.0;
  __syncthreads();

  const double* __restrict__ S = L.q.aff + 248;
  // ---- general geometry: every wave sees the same elements (lane = element)
  for (int t = lane; GEN && !RW && t < LT::NE && L.ablate != 1; t += 64) {   // K (see EARLY)
    const int lx = t % LT::EX, ly = (t / LT::EX) % LT::EY, lz = t / (LT::EX * LT::EY);
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly, gl = L.p_own0 + z0 - 1 + lz;
    if (gx < 0 || gx >= nx - 1 || gy < 0 || gy >= ny - 1 || gl < 0 || gl >= L.npl - 1) continue;
    const int n00 = gy * nx + gx;
    const double* q0 = L.xyz + (int64_t)(lat_plane(L, gl) + n00) * 3;
    const double* qz = L.xyz + (int64_t)(lat_plane(L, gl + 1) + n00) * 3;
    // K: the element's Laplacian entries were integrated once by kle_elem_laplace_kernel (element id = x + (nx-1)(y + (ny-1) layer));
    // this wave's two rows {2 part, 2 part + 1}, requested together with the corner coordinates (one memory latency, not two)
    if (!RW && t != lane) {   // (the lane's first element: requested before the accumulators were cleared)
      const double* Le = T.Lel + ((int64_t)gl * (ny - 1) + gy) * (nx - 1) + gx;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
          const int a = 2 * part + h;
          const int lo = min(a, bb), hi = max(a, bb);
          const int off = lo * 7 - (lo * (lo - 1)) / 2 + (hi - lo - 1);     // q1_off(lo, hi)
          Lv[h][bb] = a == bb ? 0.0 : Le[(int64_t)off * T.ne];
        }
    }
    double C[2][2][2][3];
    if (t != lane) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) {
            P[0][j][i][cc] = q0[(j * nx + i) * 3 + cc];
            P[1][j][i][cc] = qz[(j * nx + i) * 3 + cc];
          }
    }
    q1_haar_coeffs(P, C);
    if (RW && KLE_RW_M_OUTER) {
      switch (part) {
        case 0: kle_lat_rw_general_m<TX, TY, TZ, 0>(T, C, lx, ly, lz, z0, acc); break;
        case 1: kle_lat_rw_general_m<TX, TY, TZ, 2>(T, C, lx, ly, lz, z0, acc); break;
        case 2: kle_lat_rw_general_m<TX, TY, TZ, 4>(T, C, lx, ly, lz, z0, acc); break;
        default: kle_lat_rw_general_m<TX, TY, TZ, 6>(T, C, lx, ly, lz, z0, acc); break;
      }
      continue;
    }
    switch (part) {
      case 0: kle_lat_rows_general<TX, TY, TZ, RW, 0>(T, C, Lv, lx, ly, lz, z0, acc); break;
      case 1: kle_lat_rows_general<TX, TY, TZ, RW, 2>(T, C, Lv, lx, ly, lz, z0, acc); break;
      case 2: kle_lat_rows_general<TX, TY, TZ, RW, 4>(T, C, Lv, lx, ly, lz, z0, acc); break;
      default: kle_lat_rows_general<TX, TY, TZ, RW, 6>(T, C, Lv, lx, ly, lz, z0, acc); break;
    }
  }
  for (int t = lane; GEN && RW && t < LT::NE && L.ablate != 1; t += 64) {
    const int lx = t % LT::EX, ly = (t / LT::EX) % LT::EY, lz = t / (LT::EX * LT::EY);
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly, gl = L.p_own0 + z0 - 1 + lz;
    if (gx < 0 || gx >= nx - 1 || gy < 0 || gy >= ny - 1 || gl < 0 || gl >= L.npl - 1) continue;
    const int n00 = gy * nx + gx;
    if (!EARLY) {   // K: loads where they are needed, in locals of their own (hoisted state costs this kernel its third workgroup per CU)
      double Pl[2][2][2][3], Lvl[2][8], C[2][2][2][3];
      kle_gen_loads<RW>(T, part, n00, gx, gy, gl, Pl, Lvl);
      q1_haar_coeffs(Pl, C);
      kle_gen_dispatch<TX, TY, TZ, RW>(T, part, C, Lvl, lx, ly, lz, z0, acc);
    } else {
      double C[2][2][2][3];
      if (t != lane) kle_gen_loads<RW>(T, part, n00, gx, gy, gl, P, Lv);   // (later elements of the lane: tiles with more than 64 elements)
      q1_haar_coeffs(P, C);
      kle_gen_dispatch<TX, TY, TZ, RW>(T, part, C, Lv, lx, ly, lz, z0, acc);
    }
  }
  for (int t = lane; !GEN && t < LT::NE && L.ablate != 1; t += 64) {
    const int lx = t % LT::EX, ly = (t / LT::EX) % LT::EY, lz = t / (LT::EX * LT::EY);
    const int gx = x0 - 1 + lx, gy = y0 - 1 + ly, gl = L.p_own0 + z0 - 1 + lz;
    if (gx < 0 || gx >= nx - 1 || gy < 0 || gy >= ny - 1 || gl < 0 || gl >= L.npl - 1) continue;
    const int n00 = gy * nx + gx;
    if (t != lane) lat_affine_corners(L, n00, gl, C4);
    double E[3][3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int x = 0; x < 3; ++x) E[d][x] = C4[d + 1][x] - C4[0][x];
    double J[3][3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int x = 0; x < 3; ++x) J[d][x] = fma(S[d * 3 + 2], E[2][x], fma(S[d * 3 + 1], E[1][x], S[d * 3] * E[0][x]));
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
    switch (part) {   // wave-uniform: wave w adds the node rows {2w, 2w+1} of every element
      case 0: kle_lat_rows<TX, TY, TZ, RW, 0>(T, Ji, det, lx, ly, lz, z0, acc); break;
      case 1: kle_lat_rows<TX, TY, TZ, RW, 2>(T, Ji, det, lx, ly, lz, z0, acc); break;
      case 2: kle_lat_rows<TX, TY, TZ, RW, 4>(T, Ji, det, lx, ly, lz, z0, acc); break;
      default: kle_lat_rows<TX, TY, TZ, RW, 6>(T, Ji, det, lx, ly, lz, z0, acc); break;
    }
  }
  const int anybc = __syncthreads_or(lat_meta_commit<TX, TY, TZ, 256>(L, z0, tid, meta, rlo, zrd, nbc));

  double* __restrict__ outA = L.A;
  double* __restrict__ outR = L.Arhs;
  if (lat_tile_plain<TX, TY, TZ>(L, x0, y0, z0, zrd, anybc)) {
    // interior tile: the TX node rows of an x-line are one contiguous run of TX*243 doubles here and in HBM
    constexpr int LINE = TX * ROW, NL = TY * TZ;
    for (int l = part; l < NL; l += 4) {
      const int64_t base = (int64_t)rlo[l * TX] * 9;
      for (int i = lane; i < LINE; i += 64) {
        outA[base + i] = acc[l * LINE + i];
        if (!RW && outR && !L.rhs_clean && !L.rcrow) outR[base + i] = 0.0;   // (a compact Krhs stores no row of a tile without imposed nodes)
      }
    }
    return;
  }
  // boundary tile: one wave per scalar row (node row s, component p), CSR slot -> stencil position as in lat_store
  for (int sr = part; sr < LT::NR * 3; sr += 4) {
    const int s = sr / 3, pp = sr - s * 3;
    const int rl = rlo[s];
    if (rl < 0) continue;
    const int rx = s % TX, ry = (s / TX) % TY, rz = s / (TX * TY);
    const int x = x0 + rx, y = y0 + ry;
    const int zi = zrd[rz];
    const int cx = 3 - (x == 0) - (x == nx - 1), cy = 3 - (y == 0) - (y == ny - 1), cz = zi & 3;
    const int cc = cx * cy, len = cc * cz;
    const int bi = ((rz + 1) * LT::BY + ry + 1) * LT::BX + rx + 1;
    const bool rowbc = (nbc[bi] >> pp) & 1;
    const int64_t gbase = ((int64_t)rl * 3 + (int64_t)pp * len) * 3;
    // Krhs: the same offset in a matrix with the graph's pattern; through the row's own start in a compact one (-1: row not stored)
    int64_t rbase = gbase;
    if (!RW && outR && L.rcrow) {
      const int rr = L.rcrow[((int64_t)(z0 + rz) * ny + y) * nx + x];
      rbase = rr >= 0 ? ((int64_t)rr * 3 + (int64_t)pp * len) * 3 : -1;
    }
    for (int idx = lane; idx < len * 3; idx += 64) {
      const int k = idx / 3, q = idx - k * 3;
      const int kz = (k >= cc) + (k >= 2 * cc);
      const int rr = k - kz * cc;
      const int ky = (rr >= cx) + (rr >= 2 * cx);
      const int kx = rr - ky * cx;
      const int dz = ((zi >> (2 + 2 * kz)) & 3) - 1;
      const int dy = ky - (y != 0), dx = kx - (x != 0);
      const double v = acc[s * ROW + (pp * 27 + (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)) * 3 + q];
      double va, vr;
      if (rowbc) {
        va = vr = (!RW && q == pp && dx == 0 && dy == 0 && dz == 0) ? 1.0 : 0.0;
      } else if (!RW && ((nbc[bi + (dz * LT::BY + dy) * LT::BX + dx] >> q) & 1)) {
        va = 0.0;
        vr = -v;
      } else {
        va = v;
        vr = 0.0;
      }
      outA[gbase + idx] = va;
      if (!RW && outR && rbase >= 0) outR[rbase + idx] = vr;
    }
  }
}


// Symbolic phase of a lattice in closed form (no sort): rowptr from lat_rowptr_std, the columns of a row in
// ascending id order = z-planes in zcode order, then y, then x, clipped at the domain faces.
__global__ void lattice_symbolic_kernel(LatArgs T, int64_t n_rows, int32_t* __restrict__ rowptr, int32_t* __restrict__ colidx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_rows) return;
  if (i == n_rows) {   // one past the last row: total number of entries
    const int sx = 3 * T.nx - 2, sy = 3 * T.ny - 2;
    const bool bot = T.p_own0 == 0;
    rowptr[i] = (3 * T.n_own - (bot && T.n_own > 0) - (T.p_own0 + T.n_own == T.npl)) * sy * sx;
    return;
  }
  const int x = (int)(i % T.nx), y = (int)((i / T.nx) % T.ny), zo = (int)(i / ((int64_t)T.nx * T.ny));
  const int lo = lat_rowptr_std(T, x, y, zo);
  rowptr[i] = lo;
  const int zi = lat_zcode(T, T.p_own0 + zo);
  const int cz = zi & 3;
  int k = lo;
  for (int kz = 0; kz < cz; ++kz) {
    const int dz = ((zi >> (2 + 2 * kz)) & 3) - 1;
    const int base = lat_plane(T, T.p_own0 + zo + dz);
    for (int dy = (y == 0 ? 0 : -1); dy <= (y == T.ny - 1 ? 0 : 1); ++dy)
      for (int dx = (x == 0 ? 0 : -1); dx <= (x == T.nx - 1 ? 0 : 1); ++dx) colidx[k++] = base + (y + dy) * T.nx + x + dx;
  }
}

}  // namespace

// ---- structured topology: detection (host, once per pyn_mesh_set) and launch -------------------------
// every element of a structured Q1 block against its closed form (one thread per element; `bad` counts the mismatches)
__global__ void lattice_conn_verify_kernel(const int32_t* __restrict__ conn, const int32_t* __restrict__ P, int64_t ne, int ex, int ey, int nx,
                                           int* __restrict__ bad) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const int ix = (int)(e % ex), iy = (int)((e / ex) % ey);
  const int64_t l = e / ((int64_t)ex * ey);
  const int32_t lo = P[l] + iy * nx + ix, hi = P[l + 1] + iy * nx + ix;
  const int4 q0 = *reinterpret_cast<const int4*>(conn + e * 8), q1 = *reinterpret_cast<const int4*>(conn + e * 8 + 4);
  if (q0.x != lo || q0.y != lo + nx || q0.z != lo + nx + 1 || q0.w != lo + 1 || q1.x != hi || q1.y != hi + 1 || q1.z != hi + nx + 1 ||
      q1.w != hi + nx)
    atomicAdd(bad, 1);
}

// `at(i)`: entry i of the local connectivity (the uploaded host array, or the closed form of pyn_mesh_box); it is asked for O(layers +
// element rows) entries -- the shape guessed from them is then checked against ALL of c->d_conn on the device
int pyn_lattice_detect(pyn_ctx* c, const ConnAt& at) {
  Lattice& L = c->lat;
  (void)hipFree(L.d_P);
  (void)hipFree(L.d_zord);
  L = Lattice();
  if (c->dim != 3 || c->nn != 8 || c->n_elem < 1 || getenv("PYNAMA_NO_LATTICE")) return PYN_OK;
  const int64_t ne = c->n_elem;
  const int64_t nx = (int64_t)at(1) - at(0);
  if (nx < 2 || at(3) != at(0) + 1) return PYN_OK;
  const int64_t ex = nx - 1;
  if (ne % ex) return PYN_OK;
  // rows of elements per layer: the first element row that does not continue the bottom plane of layer 0
  int64_t ey = 0;
  const int32_t c0 = at(0);
  for (int64_t j = 0; j * ex < ne; ++j) {
    if (at(j * ex * 8) != c0 + j * nx) break;
    ey = j + 1;
  }
  if (ey < 1 || (ne / ex) % ey) return PYN_OK;
  const int64_t ny = ey + 1, ezl = ne / (ex * ey), npl = ezl + 1, nxny = nx * ny;
  if (nxny * npl != c->n_node || nxny > INT32_MAX / 2) return PYN_OK;
  std::vector<int32_t> P((size_t)npl);
  for (int64_t l = 0; l < ezl; ++l) {
    const int64_t e0 = l * ex * ey * 8;
    P[l] = at(e0);
    if (l + 1 == ezl) P[l + 1] = at(e0 + 4);
    if (l > 0 && P[l] != at((l - 1) * ex * ey * 8 + 4)) return PYN_OK;
  }
  // planes are disjoint blocks of nx*ny ids; the owned ones are consecutive in z and carry ids 0..n_owned-1
  std::vector<int32_t> sorted(P);
  std::sort(sorted.begin(), sorted.end());
  for (int64_t j = 0; j < npl; ++j)
    if (sorted[j] != j * nxny) return PYN_OK;
  if (c->n_owned % nxny) return PYN_OK;
  const int n_own = (int)(c->n_owned / nxny);
  int p0 = -1;
  for (int64_t j = 0; j < npl; ++j)
    if (P[j] == 0) p0 = (int)j;
  if (p0 < 0 || p0 + n_own > npl) return PYN_OK;
  for (int j = 0; j < n_own; ++j)
    if (P[p0 + j] != (int64_t)j * nxny) return PYN_OK;
  std::vector<int32_t> zord((size_t)npl);
  for (int64_t j = 0; j < npl; ++j) {
    int dz[3], n = 0;
    for (int d = -1; d <= 1; ++d)
      if (j + d >= 0 && j + d < npl) dz[n++] = d;
    std::sort(dz, dz + n, [&](int a, int b2) { return P[j + a] < P[j + b2]; });
    int code = n;
    for (int i = 0; i < n; ++i) code |= (dz[i] + 1) << (2 + 2 * i);
    zord[j] = code;
  }
  PYN_HIP(hipMalloc((void**)&L.d_P, npl * sizeof(int32_t)));
  PYN_HIP(hipMalloc((void**)&L.d_zord, npl * sizeof(int32_t)));
  PYN_HIP(hipMemcpy(L.d_P, P.data(), npl * sizeof(int32_t), hipMemcpyHostToDevice));
  PYN_HIP(hipMemcpy(L.d_zord, zord.data(), npl * sizeof(int32_t), hipMemcpyHostToDevice));
  {   // every element against the guessed shape
    int* d_bad = nullptr;
    int bad = 0;
    PYN_HIP(hipMalloc((void**)&d_bad, sizeof(int)));
    PYN_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), c->stream));
    lattice_conn_verify_kernel<<<(unsigned)((ne + 255) / 256), 256, 0, c->stream>>>(c->d_conn, L.d_P, ne, (int)ex, (int)ey, (int)nx, d_bad);
    PYN_HIP(hipGetLastError());
    PYN_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PYN_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(d_bad);
    if (bad) {
      (void)hipFree(L.d_P);
      (void)hipFree(L.d_zord);
      L = Lattice();
      return PYN_OK;
    }
  }
  L.nx = (int)nx;
  L.ny = (int)ny;
  L.npl = (int)npl;
  L.p_own0 = p0;
  L.n_own = n_own;
  // does the numbering have the arithmetic shape lat_plane / lat_zcode assume (one rank, or a rank's z-slab)?
  L.std_shape = true;
  for (int64_t j = 0; j < npl && L.std_shape; ++j) {
    int64_t want;
    if (j < p0) want = (n_own + j) * nxny;
    else if (j >= p0 + n_own) want = (n_own + p0 + (j - p0 - n_own)) * nxny;
    else want = (j - p0) * nxny;
    L.std_shape = P[j] == want;
  }
  L.valid = true;
  return PYN_OK;
}

template <int TX, int TY, int TZ>
static int launch_lattice(pyn_ctx* c, LatArgs& T, bool affine) {
  using LT = LatTile<TX, TY, TZ>;
  T.ntx = (T.nx + TX - 1) / TX;
  T.nty = (T.ny + TY - 1) / TY;
  const int ntz = (T.n_own + TZ - 1) / TZ;
  const int n_tiles = T.ntx * T.nty * ntz;
  static bool attr_done = false;
  if (!attr_done) {
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_q1_hex_lattice_kernel<TX, TY, TZ, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LT::BYTES));
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_q1_hex_lattice_kernel<TX, TY, TZ, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LT::BYTES));
    attr_done = true;
  }
  if (affine)
    assemble_q1_hex_lattice_kernel<TX, TY, TZ, true><<<n_tiles, TILE_THREADS, LT::BYTES, c->stream>>>(T);
  else
    assemble_q1_hex_lattice_kernel<TX, TY, TZ, false><<<n_tiles, TILE_THREADS, LT::BYTES, c->stream>>>(T);
  PYN_HIP(hipGetLastError());
  return PYN_OK;
}

// lattice descriptor -> kernel arguments (+ the one-off verification that index arithmetic may replace the loads)
static int lat_fill_args(pyn_ctx* c, LatArgs& T, double* A, double* Arhs, int* mesh_aff) {
  Lattice& L = c->lat;
  T.xyz = c->d_xyz;
  T.rowptr = c->d_rowptr;
  T.bcmask = c->d_bcmask;
  T.P = L.d_P;
  T.zord = L.d_zord;
  T.nx = L.nx;
  T.ny = L.ny;
  T.npl = L.npl;
  T.p_own0 = L.p_own0;
  T.n_own = L.n_own;
  T.ntx = T.nty = 0;
  T.bz0 = 0;
  T.bzs = 1;
  T.std_lat = 0;
  T.q = TileArgs();
  T.q.w = c->quad[0].w;
  T.q.hrs = c->quad[0].Hrs;
  T.q.hcoo = c->quad[0].HrsCoo;
  T.q.aff = getenv("PYNAMA_NO_AFFINE") ? nullptr : c->d_aff;
  T.A = A;
  T.Arhs = Arhs;
  T.dinv = nullptr;
  T.rhs_clean = c->asm_rhs_clean ? 1 : 0;
  T.rcrow = c->asm_rcrow;
  const char* ab = getenv("PYNAMA_LATTICE_ABLATE");  // diagnostics: 1 = no element phase, 4 = no plain-tile store path
  T.ablate = ab ? atoi(ab) : 0;
  T.lean = c->q1_gauss_standard && !getenv("PYNAMA_NO_LEAN") ? 1 : 0;
  PYN_TRY(pyn_mesh_all_affine(c, mesh_aff));
  if (L.std_ok < 0) {      // once per graph: may the index arithmetic replace P / zord / rowptr?
    L.std_ok = 0;
    if (L.std_shape && !getenv("PYNAMA_NO_STD_LATTICE")) {
      DevTmp flag;
      PYN_HIP(flag.alloc(sizeof(int)));
      const int one = 1;
      PYN_HIP(hipMemcpyAsync(flag.p, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
      LatArgs Tc = T;
      Tc.std_lat = 1;
      lattice_rowptr_check_kernel<<<(int)((c->n_owned + 255) / 256), 256, 0, c->stream>>>(Tc, c->d_rowptr, c->n_owned, flag.as<int>());
      int h = 0;
      PYN_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      PYN_HIP(hipStreamSynchronize(c->stream));
      L.std_ok = h;
    }
  }
  T.std_lat = L.std_ok == 1;
  return PYN_OK;
}

template <int TX, int TY, int TZ, bool GEN>
static int launch_kle_lattice(pyn_ctx* c, KleLatArgs& T, double* K, double* Krhs, double* Rw) {
  using LT = LatTile<TX, TY, TZ>;
  T.L.ntx = (T.L.nx + TX - 1) / TX;
  T.L.nty = (T.L.ny + TY - 1) / TY;
  const int n_tiles = T.L.ntx * T.L.nty * ((T.L.n_own + TZ - 1) / TZ);
  const size_t lds = (size_t)LT::NR * 243 * sizeof(double) + LT::META_INTS * sizeof(int);
  static bool attr_done = false;
  if (!attr_done) {
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_q1_hex_kle_lattice_kernel<TX, TY, TZ, false, GEN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_q1_hex_kle_lattice_kernel<TX, TY, TZ, true, GEN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  if (K) {
    if (GEN) {   // the element Laplacians, once per element
      const int64_t ne = (int64_t)(T.L.nx - 1) * (T.L.ny - 1) * (T.L.npl - 1);
      const size_t need = (size_t)28 * ne * sizeof(double);
      if (need > c->kle_lel_bytes) {
        if (c->d_kle_lel) PYN_HIP(hipFree(c->d_kle_lel));
        c->d_kle_lel = nullptr;
        c->kle_lel_bytes = 0;
        PYN_HIP(hipMalloc((void**)&c->d_kle_lel, need));
        c->kle_lel_bytes = need;
      }
      kle_elem_laplace_kernel<<<(int)((ne + 255) / 256), 256, 0, c->stream>>>(T.L, c->d_kle_lel, ne);
      T.Lel = c->d_kle_lel;
      T.ne = ne;
    }
    T.L.A = K;
    T.L.Arhs = Krhs;
    assemble_q1_hex_kle_lattice_kernel<TX, TY, TZ, false, GEN><<<n_tiles, 256, lds, c->stream>>>(T);
  }
  if (Rw) {
    T.L.A = Rw;
    T.L.Arhs = nullptr;
    assemble_q1_hex_kle_lattice_kernel<TX, TY, TZ, true, GEN><<<n_tiles, 256, lds, c->stream>>>(T);
  }
  PYN_HIP(hipGetLastError());
  return PYN_OK;
}

// KLE on lattices of parallelepipeds (the reference's box meshes): plan-free kernels
int pyn_assemble_kle_lattice(pyn_ctx* c, double alpha_d, double alpha_w, double* K, double* Krhs, double* Rw, bool* handled) {
  if (!c->lat.valid || c->quad[0].ngp != 8 || c->quad[1].ngp != 1 || getenv("PYNAMA_NO_KLE_LATTICE")) return PYN_OK;
  KleLatArgs T;
  int mesh_aff = 0;
  PYN_TRY(lat_fill_args(c, T.L, nullptr, nullptr, &mesh_aff));
  const bool affine = mesh_aff && c->aff_standard && c->aff_rw_standard && !getenv("PYNAMA_NO_AFFINE");
  // general geometry: the same plan-free four-wave kernel with the closed form of the 2x2x2 rule (standard tables only);
  // anything else falls through to the patch-plan kernels with the table-driven quadrature
  const bool general = !affine && c->q1_gauss_standard && c->q1_red_standard && !getenv("PYNAMA_NO_KLE_GENERAL");
  if (!affine && !general) return PYN_OK;
  T.alpha_d = alpha_d;
  T.alpha_w = alpha_w;
  T.wr = c->quad[1].w;
  T.hrsr = c->quad[1].Hrs;
  T.Hr = c->quad[1].H;
  T.hcoor = c->quad[1].HrsCoo;
  const char* tl = getenv("PYNAMA_KLE_LATTICE_TILE");
  if (general) {
    PYN_TRY((launch_kle_lattice<3, 3, 3, true>(c, T, K, Krhs, Rw)));
    *handled = true;
    return PYN_OK;
  }
  switch (tl ? atoi(tl) : 0) {
    case 1: PYN_TRY((launch_kle_lattice<6, 2, 2, false>(c, T, K, Krhs, Rw))); break;
    case 2: PYN_TRY((launch_kle_lattice<3, 3, 2, false>(c, T, K, Krhs, Rw))); break;
    case 3: PYN_TRY((launch_kle_lattice<4, 3, 3, false>(c, T, K, Krhs, Rw))); break;
    case 4: PYN_TRY((launch_kle_lattice<5, 2, 2, false>(c, T, K, Krhs, Rw))); break;   // 39 KB: four workgroups per CU, -3.5 % (flat 6x3x1 / 7x2x1 / 3x3x1 tiles: +0 .. +37 %)
    default: PYN_TRY((launch_kle_lattice<3, 3, 3, false>(c, T, K, Krhs, Rw))); break;
  }
  *handled = true;
  return PYN_OK;
}

int pyn_assemble_lattice(pyn_ctx* c, double* A, double* Arhs, bool* handled) {
  Lattice& L = c->lat;
  if (!L.valid || c->quad[0].ngp != 8) return PYN_OK;
  LatArgs T;
  int mesh_aff = 0;
  PYN_TRY(lat_fill_args(c, T, A, Arhs, &mesh_aff));
  T.dinv = c->asm_dinv;            // the store phases see whole rows: 1 / diagonal leaves with them (no diag_kernel pass)
  c->asm_dinv_written = T.dinv != nullptr;
  const bool affine = mesh_aff == 1 && T.q.aff != nullptr && c->aff_standard;
  // measured at 10M DOFs: parallelepipeds are store-bound -> small tiles, 5 workgroups per CU;
  // the quadrature path is FP64-bound -> 7x7x7 tiles (least redundant integration that fits the LDS twice)
  const char* tl = getenv("PYNAMA_LATTICE_TILE");
  if (!affine && c->q1_gauss_standard && T.std_lat && !tl && !getenv("PYNAMA_NO_MARCH")) {   // general geometry: FP64-bound -> z-marching kernel
    const char* mt = getenv("PYNAMA_MARCH_TILE");
    PYN_TRY(pyn_assemble_lattice_march(c, &T, mt ? atoi(mt) : 0));
    *handled = true;
    return PYN_OK;
  }
  const int sel = tl ? atoi(tl) : (affine ? 0 : 1);
  switch (sel) {
    case 1: PYN_TRY((launch_lattice<7, 7, 7>(c, T, affine))); break;
    case 2: PYN_TRY((launch_lattice<6, 6, 6>(c, T, affine))); break;
    case 3: PYN_TRY((launch_lattice<8, 6, 6>(c, T, affine))); break;
    case 4: PYN_TRY((launch_lattice<7, 6, 6>(c, T, affine))); break;
    case 5: PYN_TRY((launch_lattice<6, 6, 4>(c, T, affine))); break;
    case 6: PYN_TRY((launch_lattice<6, 5, 5>(c, T, affine))); break;
    case 7: PYN_TRY((launch_lattice<7, 4, 4>(c, T, affine))); break;
    case 8: PYN_TRY((launch_lattice<14, 3, 3>(c, T, affine))); break;
    case 9: PYN_TRY((launch_lattice<7, 5, 5>(c, T, affine))); break;
    default: PYN_TRY((launch_lattice<7, 5, 4>(c, T, affine))); break;
  }
  *handled = true;
  return PYN_OK;
}


// z-tile subset of a matrix-free product (halo / compute overlap across ranks): 0 = every tile; 1 = the tiles that read no
// ghost plane (they can run while the halo exchange is in flight); 2 = the others (bottom and / or top layer of tiles of a
// z-slab).  Returns the number of tile layers and fills T.bz0 / T.bzs.
static int mf_select_layers(LatArgs& T, int ntz, int zsel) {
  const int lo = T.p_own0 > 0 ? 1 : 0, hi = T.p_own0 + T.n_own < T.npl ? 1 : 0;   // ghost plane below / above
  T.bz0 = 0;
  T.bzs = 1;
  if (zsel == 0) return ntz;
  const int n_int = std::max(0, ntz - lo - hi);
  if (zsel == 1) {
    T.bz0 = lo;
    return n_int;
  }
  if (n_int == 0) return ntz;                    // no interior layer: the "boundary" product is the whole product
  if (lo && hi) {
    T.bzs = ntz - 1;
    return 2;
  }
  T.bz0 = hi ? ntz - 1 : 0;
  return lo + hi;
}

struct MfLaunch {   // where a (partial) product runs and where its dot partials go
  int zsel = 0, part_off = 0, max_grid = PYN_MAX_PARTIALS;
  hipStream_t st = nullptr;
};

// ---- matrix-free Laplacian (see lattice_matfree_laplace_kernel)
template <int TX, int TY, int TZ>
static int launch_matfree(pyn_ctx* c, LatArgs& T, bool affine, const double* x, double* y, bool dot, const MfLaunch& L, int* grid_out) {
  using MT = MfTile<TX, TY, TZ>;
  T.ntx = (T.nx + TX - 1) / TX;
  T.nty = (T.ny + TY - 1) / TY;
  const int n_tiles = T.ntx * T.nty * mf_select_layers(T, (T.n_own + TZ - 1) / TZ, L.zsel);
  const int grid = std::min(n_tiles, L.max_grid);   // a multiple of 8 whenever it is smaller than n_tiles: the XCD mapping survives
  if (grid_out) *grid_out = grid;
  if (grid == 0) return PYN_OK;
  const int* flag = dot ? c->d_flag : nullptr;
  double* part = dot ? c->d_part + L.part_off : nullptr;
  hipStream_t s = L.st;
  if (affine) {
    if (dot)
      lattice_matfree_laplace_kernel<TX, TY, TZ, true, true><<<grid, 256, MT::BYTES, s>>>(T, x, y, flag, part, n_tiles);
    else
      lattice_matfree_laplace_kernel<TX, TY, TZ, true, false><<<grid, 256, MT::BYTES, s>>>(T, x, y, flag, part, n_tiles);
  } else {
    if (dot)
      lattice_matfree_laplace_kernel<TX, TY, TZ, false, true><<<grid, 256, MT::BYTES, s>>>(T, x, y, flag, part, n_tiles);
    else
      lattice_matfree_laplace_kernel<TX, TY, TZ, false, false><<<grid, 256, MT::BYTES, s>>>(T, x, y, flag, part, n_tiles);
  }
  PYN_HIP(hipGetLastError());
  return PYN_OK;
}

template <int TZ>
static int launch_matfree_march(pyn_ctx* c, LatArgs& T, const double* x, double* y, bool dot, const MfLaunch& L, int* grid_out) {
  using MT = MfTile<15, 15, TZ>;
  T.ntx = (T.nx + 14) / 15;
  T.nty = (T.ny + 14) / 15;
  const int n_tiles = T.ntx * T.nty * mf_select_layers(T, (T.n_own + TZ - 1) / TZ, L.zsel);
  const int grid = std::min(n_tiles, L.max_grid);
  if (grid_out) *grid_out = grid;
  if (grid == 0) return PYN_OK;
  if (dot)
    lattice_matfree_laplace_march_kernel<TZ, true><<<grid, 256, MT::BYTES, L.st>>>(T, x, y, c->d_flag, c->d_part + L.part_off, n_tiles);
  else
    lattice_matfree_laplace_march_kernel<TZ, false><<<grid, 256, MT::BYTES, L.st>>>(T, x, y, nullptr, nullptr, n_tiles);
  PYN_HIP(hipGetLastError());
  return PYN_OK;
}

bool pyn_lattice_matfree_supported(const pyn_ctx* c) {
  return c->lat.valid && c->dim == 3 && c->nn == 8 && c->quad[0].ngp == 8;
}

// y = A x with A = the scalar Laplacian under the mask snapshot of pyn_matfree_set; x carries the ghost tail.
// zsel / part_off / max_grid / st: see MfLaunch (whole product on the context stream by default).
static int matfree_laplace_launch(pyn_ctx* c, const double* x, double* y, bool dot, const MfLaunch& L, int* grid_out) {
  PYN_CHECK(pyn_lattice_matfree_supported(c), "matrix-free operator: needs a Q1 hexahedral mesh with structured topology and the "
                                               "full-rule tables");
  PYN_CHECK(c->mf_set[PYN_MATFREE_LAPLACE], "matrix-free Laplacian: pyn_matfree_set first");
  LatArgs T;
  int mesh_aff = 0;
  PYN_TRY(lat_fill_args(c, T, nullptr, nullptr, &mesh_aff));
  T.bcmask = c->mf_mask[PYN_MATFREE_LAPLACE];
  const bool affine = mesh_aff == 1 && T.q.aff != nullptr && c->aff_standard;
  const char* tl = getenv("PYNAMA_MATFREE_TILE");
  const int sel = tl ? atoi(tl) : (affine ? 6 : 0);
  if (affine && sel >= 6) {   // parallelepipeds: column-marching kernel, 15 x 15 x TZ rows per tile
    switch (sel) {
      case 7: return launch_matfree_march<4>(c, T, x, y, dot, L, grid_out);
      case 8: return launch_matfree_march<6>(c, T, x, y, dot, L, grid_out);
      case 9: return launch_matfree_march<12>(c, T, x, y, dot, L, grid_out);
      default: return launch_matfree_march<8>(c, T, x, y, dot, L, grid_out);
    }
  }
  switch (sel) {
    case 1: PYN_TRY((launch_matfree<16, 8, 4>(c, T, affine, x, y, dot, L, grid_out))); break;
    case 2: PYN_TRY((launch_matfree<8, 8, 8>(c, T, affine, x, y, dot, L, grid_out))); break;
    case 3: PYN_TRY((launch_matfree<16, 4, 4>(c, T, affine, x, y, dot, L, grid_out))); break;
    case 4: PYN_TRY((launch_matfree<32, 4, 4>(c, T, affine, x, y, dot, L, grid_out))); break;
    case 5: PYN_TRY((launch_matfree<12, 6, 6>(c, T, affine, x, y, dot, L, grid_out))); break;
    default: PYN_TRY((launch_matfree<16, 8, 8>(c, T, affine, x, y, dot, L, grid_out))); break;
  }
  return PYN_OK;
}

template <int TX, int TY, int TZ>
static int launch_matfree_kle(pyn_ctx* c, KleLatArgs& K, bool affine, const double* x, double* y, bool dot, const MfLaunch& L, int* grid_out) {
  using MT = MfTile<TX, TY, TZ>;
  LatArgs& T = K.L;
  T.ntx = (T.nx + TX - 1) / TX;
  T.nty = (T.ny + TY - 1) / TY;
  const int n_tiles = T.ntx * T.nty * mf_select_layers(T, (T.n_own + TZ - 1) / TZ, L.zsel);
  const int grid = std::min(n_tiles, L.max_grid);
  if (grid_out) *grid_out = grid;
  if (grid == 0) return PYN_OK;
  const size_t lds = (size_t)(MT::NB + MT::NR) * 3 * sizeof(double) + ((MT::NB + 7) & ~7);
  const int* flag = dot ? c->d_flag : nullptr;
  double* part = dot ? c->d_part + L.part_off : nullptr;
  hipStream_t s = L.st;
  if (affine) {
    if (dot)
      lattice_matfree_kle_kernel<TX, TY, TZ, true, true><<<grid, 256, lds, s>>>(K, x, y, flag, part, n_tiles);
    else
      lattice_matfree_kle_kernel<TX, TY, TZ, true, false><<<grid, 256, lds, s>>>(K, x, y, flag, part, n_tiles);
  } else {
    if (dot)
      lattice_matfree_kle_kernel<TX, TY, TZ, false, true><<<grid, 256, lds, s>>>(K, x, y, flag, part, n_tiles);
    else
      lattice_matfree_kle_kernel<TX, TY, TZ, false, false><<<grid, 256, lds, s>>>(K, x, y, flag, part, n_tiles);
  }
  PYN_HIP(hipGetLastError());
  return PYN_OK;
}

// y = K x with K = the KLE stiffness under the per-DOF mask snapshot (pyn_matfree_set supplied alpha_d, alpha_w and took the mask)
static int matfree_kle_launch(pyn_ctx* c, const double* x, double* y, bool dot, const MfLaunch& L, int* grid_out) {
  PYN_CHECK(pyn_lattice_matfree_supported(c) && c->quad[1].ngp == 1,
            "matrix-free operator: needs a Q1 hexahedral mesh with structured topology and the full- and reduced-rule tables");
  PYN_CHECK(c->mf_set[PYN_MATFREE_KLE], "matrix-free KLE operator: pyn_matfree_set first");
  KleLatArgs K;
  int mesh_aff = 0;
  PYN_TRY(lat_fill_args(c, K.L, nullptr, nullptr, &mesh_aff));
  K.L.bcmask = c->mf_mask[PYN_MATFREE_KLE];
  K.alpha_d = c->mf_alpha_d;
  K.alpha_w = c->mf_alpha_w;
  K.wr = c->quad[1].w;
  K.hrsr = c->quad[1].Hrs;
  K.Hr = c->quad[1].H;
  K.hcoor = c->quad[1].HrsCoo;
  const bool affine = mesh_aff == 1 && K.L.q.aff != nullptr && c->aff_standard;
  const char* tl = getenv("PYNAMA_MATFREE_TILE");
  switch (tl ? atoi(tl) : 0) {
    case 1: PYN_TRY((launch_matfree_kle<16, 8, 4>(c, K, affine, x, y, dot, L, grid_out))); break;
    case 2: PYN_TRY((launch_matfree_kle<6, 6, 6>(c, K, affine, x, y, dot, L, grid_out))); break;
    case 3: PYN_TRY((launch_matfree_kle<16, 4, 4>(c, K, affine, x, y, dot, L, grid_out))); break;
    default: PYN_TRY((launch_matfree_kle<8, 8, 8>(c, K, affine, x, y, dot, L, grid_out))); break;
  }
  return PYN_OK;
}

int pyn_lattice_matfree_spmv(pyn_ctx* c, const double* x, double* y, bool dot, int* grid_out) {
  MfLaunch L;
  L.st = c->stream;
  return matfree_laplace_launch(c, x, y, dot, L, grid_out);
}

int pyn_lattice_matfree_kle_spmv(pyn_ctx* c, const double* x, double* y, bool dot, int* grid_out) {
  MfLaunch L;
  L.st = c->stream;
  return matfree_kle_launch(c, x, y, dot, L, grid_out);
}

// part of a product (op = PYN_MATFREE_*): zsel 1 = the tiles that read no ghost plane, 2 = the others (see mf_select_layers)
int pyn_lattice_matfree_part(pyn_ctx* c, int op, const double* x, double* y, bool dot, int zsel, int part_off, int max_grid, hipStream_t st,
                             int* grid_out) {
  MfLaunch L;
  L.zsel = zsel;
  L.part_off = part_off;
  L.max_grid = max_grid;
  L.st = st;
  return op == PYN_MATFREE_KLE ? matfree_kle_launch(c, x, y, dot, L, grid_out) : matfree_laplace_launch(c, x, y, dot, L, grid_out);
}

// Node graph of a lattice whose numbering has the arithmetic shape (one rank, or a rank's z-slab): built directly,
// without the sort of pyn_csr_symbolic.  *done = false when the mesh does not qualify.
int pyn_lattice_symbolic(pyn_ctx* c, bool* done) {
  *done = false;
  const Lattice& L = c->lat;
  if (!L.valid || !L.std_shape || getenv("PYNAMA_NO_LATTICE_SYMBOLIC")) return PYN_OK;
  LatArgs T;
  T.xyz = c->d_xyz;
  T.rowptr = nullptr;
  T.bcmask = nullptr;
  T.P = L.d_P;
  T.zord = L.d_zord;
  T.nx = L.nx;
  T.ny = L.ny;
  T.npl = L.npl;
  T.p_own0 = L.p_own0;
  T.n_own = L.n_own;
  T.ntx = T.nty = 0;
  T.std_lat = 1;
  T.ablate = 0;
  T.lean = 0;
  T.q = TileArgs();
  T.A = T.Arhs = nullptr;
  const int64_t sx = 3 * (int64_t)L.nx - 2, sy = 3 * (int64_t)L.ny - 2;
  const int64_t nnz = (3 * (int64_t)L.n_own - (L.p_own0 == 0 ? 1 : 0) - (L.p_own0 + L.n_own == L.npl ? 1 : 0)) * sy * sx;
  PYN_CHECK(nnz > 0 && nnz < (int64_t)INT32_MAX, "pattern has %lld entries (int32 CSR limit)", (long long)nnz);
  (void)hipFree(c->d_rowptr);
  (void)hipFree(c->d_colidx);
  c->d_rowptr = nullptr;
  c->d_colidx = nullptr;
  PYN_HIP(hipMalloc((void**)&c->d_rowptr, (c->n_owned + 1) * sizeof(int32_t)));
  PYN_HIP(hipMalloc((void**)&c->d_colidx, nnz * sizeof(int32_t)));
  lattice_symbolic_kernel<<<(int)((c->n_owned + 1 + 255) / 256), 256, 0, c->stream>>>(T, c->n_owned, c->d_rowptr, c->d_colidx);
  PYN_HIP(hipGetLastError());
  c->nnzb = nnz;
  *done = true;
  return PYN_OK;
}
