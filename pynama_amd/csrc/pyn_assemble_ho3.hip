// Second-order elements (ngl = 3: 9-node quadrilaterals, 27-node hexahedra) on structured meshes: atomics-free assembly of the KLE
// matrices -- the element order every case of the reference runs (src/cases/*.yaml: `ngl: 3`; Spectral picks Gauss(3)^dim and
// Gauss(2)^dim there, src/elements/spectral.py:41-43).
//
// Reference: Spectral.getElemKLEMatrices (src/elements/spectral.py:89-157) per cell inside FreeSlip.buildKLEMats
// (src/cases/base_problem.py:499-552), Mat.setIndices2One (src/matrices/mat_generator.py:113-118).
//
// Why not the LDS tiles of the Q1 kernels: a vertex row of the 3-D K has 125 neighbours x 9 values = 9 KB, a tile of rows does
// not fit the LDS.  Instead a workgroup owns a RUN of consecutive node rows of one x-line -- one contiguous piece of every
// block-CSR value array -- keeps exactly that piece in LDS, and enumerates the (row, element, column node) triples that feed it:
// 729 (3-D) / 81 (2-D) per element, each computed by ONE lane of ONE workgroup, so nothing is integrated twice.  Rows of a line
// alternate between two classes (even / odd x), lines come in 2^(dim-1) classes (parity of y, z): one launch per line class, its
// LDS sized for that class.  `ds_add_f64` sums the <= 8 element contributions of an entry; the piece leaves as one coalesced
// copy, with the Dirichlet routing K / Krhs / unit diagonal applied on the way out in runs that touch an imposed DOF.
//
// Element matrices: on a parallelogram / parallelepiped J is constant, so
//   sum_g w_g detJ (J^-1 Hrs_g)^T (J^-1 Hrs_g) = detJ J^-1 [ sum_g w_g Hrs_g Hrs_g^T ] J^-T     (an identity, any rule)
// and every block is a contraction of the per-element J^-1 (pre-pass, 80 B per element) with reference matrices computed ONCE
// from the uploaded tables (pyn_ho3_tables):  with Y_pq = sum_rs Ji_pr Ji_qs Tr_rs[a][b],  Q = Ji^T Ji,
//   K [(a,p),(b,q)] = detJ ( alpha_d Y_pq - alpha_w Y_qp ) + d_pq detJ ( sum_rs Q_rs Tf_rs[a][b] + alpha_w tr Y )
//   Rw[(a,p),(b,k)] = detJ sum_{(p,k,d,s) in curl_w} s Ji_d. Uf_.[a][b] + alpha_w detJ sum_{(k,p,d,s) in curl_v} s Ji_d. Ur_.[b][a]
// (spectral.py:124-156 written out; checked against the oracle's quadrature loop).  Meshes with a non-affine element keep the
// generic workgroup-per-element kernel.
//
// The same row-run scheme serves FIRST-order lattices (NGL = 2) where no dedicated kernel exists: 2-D Q1 quadrilaterals (K, Krhs, Rw,
// scalar Laplacian) and, on every lattice (2-D / 3-D, ngl 2 / 3), the first-order operators SrT / DivSrT / Curl of
// Spectral.getElemKLEOperators (src/elements/spectral.py:159-218; Operators.setValues, src/matrices/mat_generator.py:157-170): at the
// nodal rule  M[(a,p),(b,q)] = detJ sum_t [row_t = p, col_t = q] coef_t sum_x Ji[der_t][x] Un_x[a][b],  Un_x[a][b] = sum_g w H[a] Hrs_x[b].
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>

#include "pyn_internal.h"

namespace {

// lattice offset (x, y[, z]) in {0, 1, 2} of local node a: the reference's vertex / edge / face / interior order
// (spectral.py:346-431, fixture tests/golden/g2_tables.npz `order_*`); 2-D carries the x ~ -r, y ~ -s flip of SURVEY.md A.2
constexpr int LOC2[9][2] = {{0, 0}, {2, 0}, {2, 2}, {0, 2}, {1, 0}, {2, 1}, {1, 2}, {0, 1}, {1, 1}};
// first-order cells: the corners in DMPlex closure order (src/tests/test_domain.py:26-30, 94-104)
constexpr int LOQ2[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
constexpr int LOQ3[8][3] = {{0, 0, 0}, {0, 1, 0}, {1, 1, 0}, {1, 0, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
constexpr int LOC3[27][3] = {{0, 0, 0}, {0, 2, 0}, {2, 2, 0}, {2, 0, 0}, {0, 0, 2}, {2, 0, 2}, {2, 2, 2}, {0, 2, 2}, {0, 1, 0},
                             {1, 2, 0}, {2, 1, 0}, {1, 0, 0}, {1, 0, 2}, {2, 1, 2}, {1, 2, 2}, {0, 1, 2}, {2, 0, 1}, {0, 0, 1},
                             {0, 2, 1}, {2, 2, 1}, {1, 1, 0}, {1, 1, 2}, {1, 0, 1}, {1, 2, 1}, {2, 1, 1}, {0, 1, 1}, {1, 1, 1}};

inline int loc_of(int dim, int ngl, int a, int d) {
  if (ngl == 2) return dim == 2 ? LOQ2[a][d] : LOQ3[a][d];
  return dim == 2 ? LOC2[a][d] : LOC3[a][d];
}
inline int tens_of(int dim, int ngl, int a) {
  int t = 0;
  for (int d = dim - 1; d >= 0; --d) t = t * ngl + loc_of(dim, ngl, a, d);
  return t;
}

enum { M_K = 0, M_RW = 1, M_LAP = 2, M_OP = 3 };
constexpr int MAX_TERMS = 32;

struct Ho3Args {
  int EX, EY, EZ, NX, NY, npl, p_own0, n_own;
  const int32_t* P;
  const int32_t* rowptr;
  const uint8_t* nbits;   // per local node: bit p = DOF p imposed; null = nothing imposed
  const uint8_t* runflag; // per owned node that starts a run: OR of nbits over the run's node box (ho3_runflag_kernel); null with nbits
  const double* geom;     // [n_elem][GS]
  const double* tabs;
  const double* tabs1d;   // [8][NGL][NGL] 1-D factors (Mf Df Sf Mr Dr Sr Mn Dn) when the records are verified tensor products, else null
  double alpha_d, alpha_w;
  double* A;
  double* Arhs;
  int rhs_clean;          // Arhs holds zeros wherever this Dirichlet set leaves zeros: runs without an imposed DOF skip it
  const int32_t* rcrow;   // Arhs is a COMPACT imposed-column matrix (pyn_rhs.hip): first block of every owned node row in it, -1 = not stored
  int par_y, par_z;       // class of the x-lines of this launch (parity of the local y / z index)
  int nruns, nly;         // runs per x-line, lines of this class per plane (3-D)
  int nwork;              // runs of this launch (the workgroups walk them)
  int so0;                // first owned plane (3-D) / line (2-D), as an owned index, whose local index has the parity of the class
  int img_len;            // doubles of LDS behind the kernel
  int pstd;               // P[j] follows the slab numbering's closed form
  int diag;               // every element's J^-1 is diagonal (axis-aligned boxes): the blocks' diagonal forms apply
  int ablate;             // diagnostics (PYNAMA_HO3_ABLATE): 1 no unit loop (zero + copy out only), 2 no copy out, 4 no LDS adds; 8 one double per store,
                          // 16 no run is flagged (no Dirichlet bits are read: WRONG matrices, timing only), 32 one geometry address for all elements
  int step;               // distance (owned index) between two lines of a class along the slow axis: 2 (ngl 3), 1 (ngl 2)
  // first-order operator form (M_OP): block shape and the (row component, column component, derivative axis, coefficient) terms
  int obr, obc, nterms;
  int t_row[MAX_TERMS], t_col[MAX_TERMS], t_der[MAX_TERMS];
  double t_coef[MAX_TERMS];
};

template <int NGL>
__device__ __forceinline__ void axis_range(int c, int N, int& lo, int& n) {
  const int h = NGL == 2 ? 1 : ((c & 1) ? 1 : 2);
  lo = max(0, c - h);
  n = min(N - 1, c + h) - lo + 1;
}

// sum of the x-extents n_x(x') of the rows x' < x of an x-line (n_x as axis_range gives it): the row offsets inside a line in closed form
template <int NGL>
__device__ __forceinline__ int x_prefix(int x, int NX) {
  if (x <= 0) return 0;
  if (NGL == 2) return 2 + 3 * (x - 1) - (x == NX ? 1 : 0);
  return 3 + 3 * (x >> 1) + 5 * ((x - 1) >> 1) - (x == NX ? 2 : 0);
}

// exact k / n for 0 <= k < 2048, n in {2, 3, 4, 5, 6, 9, 15, 25}
__device__ __forceinline__ int small_div(int k, int n) { return (k * (65536 / n + 1)) >> 16; }

// corner cn of an element (first 2^dim local nodes) as lattice bits x | y << 1 | z << 2 -- LOC2 / LOC3 halved, spelled out for device code
constexpr int CB2[4] = {0, 1, 3, 2};
constexpr int CB3[8] = {0, 2, 3, 1, 4, 5, 7, 6};
constexpr bool corner_bits_match() {
  for (int cn = 0; cn < 4; ++cn)
    if (CB2[cn] != ((LOC2[cn][0] >> 1) | ((LOC2[cn][1] >> 1) << 1))) return false;
  for (int cn = 0; cn < 8; ++cn)
    if (CB3[cn] != ((LOC3[cn][0] >> 1) | ((LOC3[cn][1] >> 1) << 1) | ((LOC3[cn][2] >> 1) << 2))) return false;
  return true;
}
static_assert(corner_bits_match(), "corner tables out of step with the local node order");
constexpr bool q1_corners_match() {
  for (int cn = 0; cn < 4; ++cn)
    if (CB2[cn] != (LOQ2[cn][0] | (LOQ2[cn][1] << 1))) return false;
  for (int cn = 0; cn < 8; ++cn)
    if (CB3[cn] != (LOQ3[cn][0] | (LOQ3[cn][1] << 1) | (LOQ3[cn][2] << 2))) return false;
  return true;
}
static_assert(q1_corners_match(), "first-order corner table out of step");

template <int DIM>
__global__ void __launch_bounds__(256) ho3_geom_kernel(const int32_t* __restrict__ conn, const double* __restrict__ xyz, int64_t n_elem, int NN,
                                                       const double* __restrict__ hcoo, double* __restrict__ geom, int* __restrict__ not_affine) {
  constexpr int NC = 1 << DIM, GS = DIM == 3 ? 10 : 6;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_elem) return;
  double X[NC][DIM];
#pragma unroll
  for (int cn = 0; cn < NC; ++cn) {
    const double* p = xyz + (int64_t)conn[e * NN + cn] * DIM;
#pragma unroll
    for (int x = 0; x < DIM; ++x) X[cn][x] = p[x];
  }
  // J = HrsCoo . X at the first point of the full rule (spectral.py:120); constant on a parallelepiped
  double J[DIM * DIM], Ji[DIM * DIM];
#pragma unroll
  for (int r = 0; r < DIM; ++r)
#pragma unroll
    for (int x = 0; x < DIM; ++x) {
      double s = 0.0;
#pragma unroll
      for (int cn = 0; cn < NC; ++cn) s = fma(hcoo[r * NC + cn], X[cn][x], s);
      J[r * DIM + x] = s;
    }
  double det;
  if constexpr (DIM == 2) {
    det = J[0] * J[3] - J[1] * J[2];
    const double rr = 1.0 / det;
    Ji[0] = J[3] * rr;
    Ji[1] = -J[1] * rr;
    Ji[2] = -J[2] * rr;
    Ji[3] = J[0] * rr;
  } else {
    const double c00 = J[4] * J[8] - J[5] * J[7], c01 = J[5] * J[6] - J[3] * J[8], c02 = J[3] * J[7] - J[4] * J[6];
    det = J[0] * c00 + J[1] * c01 + J[2] * c02;
    const double rr = 1.0 / det;
    Ji[0] = c00 * rr;
    Ji[1] = (J[2] * J[7] - J[1] * J[8]) * rr;
    Ji[2] = (J[1] * J[5] - J[2] * J[4]) * rr;
    Ji[3] = c01 * rr;
    Ji[4] = (J[0] * J[8] - J[2] * J[6]) * rr;
    Ji[5] = (J[2] * J[3] - J[0] * J[5]) * rr;
    Ji[6] = c02 * rr;
    Ji[7] = (J[1] * J[6] - J[0] * J[7]) * rr;
    Ji[8] = (J[0] * J[4] - J[1] * J[3]) * rr;
  }
  double* g = geom + e * GS;
#pragma unroll
  for (int i = 0; i < DIM * DIM; ++i) g[i] = Ji[i];
  g[DIM * DIM] = det;
  if constexpr (DIM == 2) g[5] = 0.0;
  if (not_affine) {
    // corner cn sits at the lattice offsets LOC[cn] / 2: a parallelepiped is X_o + sum_d bit_d (X_d - X_o)
    constexpr int cb2[4] = {0, 1, 3, 2}, cb3[8] = {0, 2, 3, 1, 4, 5, 7, 6};   // = CB2 / CB3
    int o = 0, ax[DIM];
    int bits[NC];
#pragma unroll
    for (int cn = 0; cn < NC; ++cn) bits[cn] = DIM == 2 ? cb2[cn & 3] : cb3[cn];
#pragma unroll
    for (int cn = 0; cn < NC; ++cn) {
      if (bits[cn] == 0) o = cn;
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        if (bits[cn] == (1 << d)) ax[d] = cn;
    }
    double na = 0.0, h2 = 0.0;
#pragma unroll
    for (int cn = 0; cn < NC; ++cn)
#pragma unroll
      for (int x = 0; x < DIM; ++x) {
        double pr = X[o][x];
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          if ((bits[cn] >> d) & 1) pr += X[ax[d]][x] - X[o][x];
        const double df = X[cn][x] - pr;
        na = fma(df, df, na);
        if (bits[cn] == NC - 1) h2 = fma(X[cn][x] - X[o][x], X[cn][x] - X[o][x], h2);
      }
    if (!(na <= 1e-25 * h2) || !(det > 0.0)) *not_affine = 1;   // round-off of the coordinates only (as element_is_affine, pyn_q1_hex.h)
    // axis-aligned boxes (every box mesh the reference makes, src/domain/dmplex.py:16-21): J is diagonal up to the round-off of its
    // cancelling sums; the kernels then take the diagonal forms of the blocks
    double dmax = 0.0, omax = 0.0;
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
      for (int x = 0; x < DIM; ++x) {
        if (r == x) dmax = fmax(dmax, fabs(J[r * DIM + x]));
        else omax = fmax(omax, fabs(J[r * DIM + x]));
      }
    if (!(omax <= 1e-14 * dmax)) not_affine[1] = 1;
  }
}

__global__ void ho3_pack_bits_kernel(const uint8_t* __restrict__ mask, int64_t n_node, int ndof, uint8_t* __restrict__ bits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_node) return;
  int m = 0;
  for (int p = 0; p < ndof; ++p) m |= (mask[i * ndof + p] ? 1 : 0) << p;
  bits[i] = (uint8_t)m;
}

// Reference matrices, one contiguous RECORD per node pair (a, b) in tensor order (TS doubles, 16-byte aligned):
//   [0, DD)           Tf_rs[a][b] = sum_g w Hrs_r[a] Hrs_s[b]   (full rule)
//   [DD, 2 DD)        Tr_rs[a][b]                                (reduced rule)
//   [2 DD, +DIM)      Uf_x[a][b]  = sum_g w H[a] Hrs_x[b]        (full rule)
//   [.., +DIM)        Ur_x[b][a]                                 (reduced rule, transposed: what Rw's (a, b) block reads)
//   [.., +DIM)        Un_x[a][b]                                 (nodal rule: the first-order operators)
// A lane reads its pair's record with one base address and immediate offsets (vector loads), lanes of a group (consecutive b) read
// consecutive records.
template <int DIM>
struct TabRec {
  static constexpr int DD = DIM * DIM;
  static constexpr int TS = (2 * DD + 3 * DIM + 1) & ~1;
  static constexpr int TF = 0, TR = DD, UF = 2 * DD, URT = 2 * DD + DIM, UN = 2 * DD + 2 * DIM;
};

// The record of node pair (a, b) rebuilt from the 1-D factors of a tensor-product element (LDS, no global load in the unit loop):
// with m_d = M[a_d][b_d], dab_d = D[a_d][b_d] = sum w h'_a h_b, dba_d = D[b_d][a_d], s_d = S[a_d][b_d] per axis d,
//   T_rs = prod_d ( d == r == s ? s_d : d == r ? dab_d : d == s ? dba_d : m_d ),   U_x[a][b] = dba_x prod_{d != x} m_d,
//   U_x[b][a] = dab_x prod_{d != x} m_d (M symmetric)
template <int DIM, int NGL, int MAT>
__device__ __forceinline__ void ho3_record_1d(const double* __restrict__ t1, const int (&la)[3], const int (&lb)[3], double* __restrict__ rec) {
  using TR_ = TabRec<DIM>;
  constexpr int N1 = NGL * NGL;
  auto fac = [&](int tab, int d, bool tr) { return t1[tab * N1 + (tr ? lb[d] * NGL + la[d] : la[d] * NGL + lb[d])]; };
  if (MAT == M_K || MAT == M_LAP) {
#pragma unroll
    for (int rule = 0; rule < (MAT == M_K ? 2 : 1); ++rule) {
      double m[DIM], dab[DIM], dba[DIM], sd[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        m[d] = fac(3 * rule + 0, d, false);
        dab[d] = fac(3 * rule + 1, d, false);
        dba[d] = fac(3 * rule + 1, d, true);
        sd[d] = fac(3 * rule + 2, d, false);
      }
#pragma unroll
      for (int r = 0; r < DIM; ++r)
#pragma unroll
        for (int s2 = 0; s2 < DIM; ++s2) {
          double p = 1.0;
#pragma unroll
          for (int d = 0; d < DIM; ++d) p *= (d == r && d == s2) ? sd[d] : (d == r ? dab[d] : (d == s2 ? dba[d] : m[d]));
          rec[(rule == 0 ? TR_::TF : TR_::TR) + r * DIM + s2] = p;
        }
    }
  } else if (MAT == M_RW) {
    double mf[DIM], mr[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      mf[d] = fac(0, d, false);
      mr[d] = fac(3, d, false);
    }
#pragma unroll
    for (int x = 0; x < DIM; ++x) {
      double pf = fac(1, x, true), pr = fac(4, x, false);
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        if (d != x) {
          pf *= mf[d];
          pr *= mr[d];
        }
      rec[TR_::UF + x] = pf;
      rec[TR_::URT + x] = pr;
    }
  } else {
    double mn[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) mn[d] = fac(6, d, false);
#pragma unroll
    for (int x = 0; x < DIM; ++x) {
      double pn = fac(7, x, true);
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        if (d != x) pn *= mn[d];
      rec[TR_::UN + x] = pn;
    }
  }
}

// The same block for an element whose J^-1 is DIAGONAL (axis-aligned boxes), straight from the 1-D factors: with j_d = Ji[d][d],
//   Y_pq = j_p j_q Tr_pq,  Tr_pp = sR_p prod_{d != p} mR_d,  Tr_pq = dabR_p dbaR_q mR_t (t the third axis),  lap = sum_r j_r^2 Tf_rr,
// i.e. ~65 FP64 operations per 3-D K block instead of ~280 (the dense J^-1 Tr J^-T products and the full 18-entry record).
template <int DIM, int NGL, int MAT>
__device__ __forceinline__ void ho3_block_diag(const double* __restrict__ g, const double* __restrict__ t1, const int (&la)[3], const int (&lb)[3],
                                               double alpha_d, double alpha_w, double (&v)[3][3]) {
  constexpr int DD = DIM * DIM, N1 = NGL * NGL;
  auto fac = [&](int tab, int d, bool tr) { return t1[tab * N1 + (tr ? lb[d] * NGL + la[d] : la[d] * NGL + lb[d])]; };
  double j[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) j[d] = g[d * DIM + d];
  const double det = g[DD];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 3; ++q) v[p][q] = 0.0;
  // prod_{d != x} m_d for the mass factors of rule `tab0 / 3`
  auto others = [&](int tab0, double (&po)[DIM], double (&m)[DIM]) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) m[d] = fac(tab0, d, false);
    if (DIM == 2) {
      po[0] = m[1];
      po[1] = m[0];
    } else {
      po[0] = m[1] * m[DIM - 1];
      po[1] = m[0] * m[DIM - 1];
      po[DIM - 1] = m[0] * m[1];
    }
  };
  if (MAT == M_OP) {   // v[0][d] = detJ j_d Un_d[a][b]
    double po[DIM], m[DIM];
    others(6, po, m);
#pragma unroll
    for (int d = 0; d < DIM; ++d) v[0][d] = det * j[d] * fac(7, d, true) * po[d];
    return;
  }
  if (MAT == M_RW) {   // gU_d = detJ j_d Uf_d[a][b], gR_d = alpha_w detJ j_d Ur_d[b][a]
    double pf[DIM], pr[DIM], m[DIM];
    others(0, pf, m);
    others(3, pr, m);
    double gU[DIM], gR[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      const double dj = det * j[d];
      gU[d] = dj * fac(1, d, true) * pf[d];
      gR[d] = alpha_w * dj * fac(4, d, false) * pr[d];
    }
    if (DIM == 2) {
      v[0][0] = gU[1] - gR[1];
      v[1][0] = -gU[0] + gR[0];
    } else {
      constexpr int CURL3[6][4] = {{0, 2, 1, 1}, {0, 1, 2, -1}, {1, 0, 2, 1}, {1, 2, 0, -1}, {2, 1, 0, 1}, {2, 0, 1, -1}};
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int r = CURL3[i][0], cp = CURL3[i][1], d = CURL3[i][2];
        const double sg = (double)CURL3[i][3];
        const double u = d == 0 ? gU[0] : (d == 1 ? gU[1] : gU[DIM - 1]);
        const double w = d == 0 ? gR[0] : (d == 1 ? gR[1] : gR[DIM - 1]);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            if (p == r && q == cp) v[p][q] += sg * u;
            if (p == cp && q == r) v[p][q] += sg * w;
          }
      }
    }
    return;
  }
  double jj[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) jj[d] = j[d] * j[d];
  double lap = 0.0;
  {
    double po[DIM], m[DIM];
    others(0, po, m);
#pragma unroll
    for (int r = 0; r < DIM; ++r) lap = fma(jj[r], fac(2, r, false) * po[r], lap);
  }
  if (MAT == M_LAP) {
    v[0][0] = det * lap;
    return;
  }
  double po[DIM], m[DIM], dab[DIM], dba[DIM];
  others(3, po, m);
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    dab[d] = fac(4, d, false);
    dba[d] = fac(4, d, true);
  }
  double trd[DIM], trY = 0.0;
#pragma unroll
  for (int p = 0; p < DIM; ++p) {
    trd[p] = fac(5, p, false) * po[p];
    trY = fma(jj[p], trd[p], trY);
  }
  const double dg = det * fma(alpha_w, trY, lap);
  const double cd = det * (alpha_d - alpha_w);
#pragma unroll
  for (int p = 0; p < DIM; ++p) v[p][p] = fma(cd * jj[p], trd[p], dg);
#pragma unroll
  for (int p = 0; p < DIM; ++p)
#pragma unroll
    for (int q = p + 1; q < DIM; ++q) {
      const double mt = DIM == 2 ? 1.0 : m[3 - p - q];     // the axis that is neither p nor q
      const double c = det * j[p] * j[q] * mt;
      const double tpq = dab[p] * dba[q], tqp = dab[q] * dba[p];   // Tr_pq / m_t, Tr_qp / m_t
      v[p][q] = c * (alpha_d * tpq - alpha_w * tqp);
      v[q][p] = c * (alpha_d * tqp - alpha_w * tpq);
    }
}

// one block of the element matrix of element `g` (J^-1, detJ) for the node pair whose table record is `tb`
template <int DIM, int MAT>
__device__ __forceinline__ void ho3_block(const double* __restrict__ g, const double* __restrict__ tb, double alpha_d, double alpha_w,
                                          double (&v)[3][3]) {
  using TR_ = TabRec<DIM>;
  constexpr int DD = DIM * DIM;
  double Ji[DIM][DIM];
#pragma unroll
  for (int x = 0; x < DIM; ++x)
#pragma unroll
    for (int r = 0; r < DIM; ++r) Ji[x][r] = g[x * DIM + r];
  const double det = g[DD];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 3; ++q) v[p][q] = 0.0;
  if (MAT == M_OP) {   // v[0][d] = detJ sum_x Ji[d][x] Un_x[a][b]: H_a grad_d N_b at the nodal rule; the caller places the terms
    double un[DIM];
#pragma unroll
    for (int x = 0; x < DIM; ++x) un[x] = tb[TR_::UN + x];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      double s0 = 0.0;
#pragma unroll
      for (int x = 0; x < DIM; ++x) s0 = fma(Ji[d][x], un[x], s0);
      v[0][d] = det * s0;
    }
    return;
  }
  if (MAT == M_K || MAT == M_LAP) {
    double tf[DIM][DIM];
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
      for (int s = 0; s < DIM; ++s) tf[r][s] = tb[TR_::TF + r * DIM + s];
    double lap = 0.0;   // sum_rs Q_rs Tf_rs, Q = Ji^T Ji
#pragma unroll
    for (int x = 0; x < DIM; ++x)
#pragma unroll
      for (int r = 0; r < DIM; ++r) {
        double t = 0.0;
#pragma unroll
        for (int s = 0; s < DIM; ++s) t = fma(tf[r][s], Ji[x][s], t);
        lap = fma(Ji[x][r], t, lap);
      }
    if (MAT == M_LAP) {
      v[0][0] = det * lap;
      return;
    }
    double tr[DIM][DIM];
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
      for (int s = 0; s < DIM; ++s) tr[r][s] = tb[TR_::TR + r * DIM + s];
    double X[DIM][DIM], Y[DIM][DIM];
#pragma unroll
    for (int p = 0; p < DIM; ++p)
#pragma unroll
      for (int s = 0; s < DIM; ++s) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < DIM; ++r) t = fma(Ji[p][r], tr[r][s], t);
        X[p][s] = t;
      }
    double trY = 0.0;
#pragma unroll
    for (int p = 0; p < DIM; ++p)
#pragma unroll
      for (int q = 0; q < DIM; ++q) {
        double t = 0.0;
#pragma unroll
        for (int s = 0; s < DIM; ++s) t = fma(X[p][s], Ji[q][s], t);
        Y[p][q] = t;
        if (p == q) trY += t;
      }
    const double dg = det * fma(alpha_w, trY, lap);
#pragma unroll
    for (int p = 0; p < DIM; ++p)
#pragma unroll
      for (int q = 0; q < DIM; ++q) v[p][q] = det * (alpha_d * Y[p][q] - alpha_w * Y[q][p]) + (p == q ? dg : 0.0);
  } else {
    // Rw: gU[d] = sum_x Ji[d][x] Uf_x[a][b] (H_a grad_d N_b, full rule), gR[d] = sum_x Ji[d][x] Ur_x[b][a] (grad_d N_a H_b, reduced rule)
    double gU[DIM], gR[DIM];
    double uf[DIM], ur[DIM];
#pragma unroll
    for (int x = 0; x < DIM; ++x) {
      uf[x] = tb[TR_::UF + x];
      ur[x] = tb[TR_::URT + x];
    }
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int x = 0; x < DIM; ++x) {
        s0 = fma(Ji[d][x], uf[x], s0);
        s1 = fma(Ji[d][x], ur[x], s1);
      }
      gU[d] = det * s0;
      gR[d] = alpha_w * det * s1;
    }
    if (DIM == 2) {
      // curl_w = [(0,0,1,+), (1,0,0,-)], curl_v = [(0,1,0,+), (0,0,1,-)]   (indWCurl / indCurl, spectral.py:26-27)
      v[0][0] = gU[1] - gR[1];
      v[1][0] = -gU[0] + gR[0];
    } else {
      // (curl w)_r = s d_d w_comp and (curl v)_r = s d_d v_comp as (r, comp, d, s): indWCurl = indCurl (spectral.py:29-32) with the
      // alternating sign of :128-129, 147-148
      constexpr int CURL3[6][4] = {{0, 2, 1, 1}, {0, 1, 2, -1}, {1, 0, 2, 1}, {1, 2, 0, -1}, {2, 1, 0, 1}, {2, 0, 1, -1}};
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int r = CURL3[i][0], cp = CURL3[i][1], d = CURL3[i][2];
        const double s = (double)CURL3[i][3];
        const double u = d == 0 ? gU[0] : (d == 1 ? gU[1] : gU[DIM - 1]);
        const double w = d == 0 ? gR[0] : (d == 1 ? gR[1] : gR[DIM - 1]);
        // full: row component r, column component cp; reduced: row component cp, column component r
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            if (p == r && q == cp) v[p][q] += s * u;
            if (p == cp && q == r) v[p][q] += s * w;
          }
      }
    }
  }
}

// A workgroup walks runs w = blockIdx.x, blockIdx.x + gridDim.x, ... of one class of x-lines.  What a run needs from memory before it can
// start (row offsets, Dirichlet bits of its node box, J^-1 / detJ of its elements, first ids of its planes) is requested one run AHEAD,
// right after the barrier that opens the triple loop of the current run, and parked in the second set of small LDS arrays when that
// loop is over: the loads have a whole run to arrive, nothing waits for the stores of the run before (they drain while the next run
// is zeroed and integrated), and a run's own phases no longer depend on four other workgroups of the CU to be overlapped.
struct Ho3Run {
  int x0, nrows, cy, cz, ylo, n_y, zlo, n_z;
  int64_t row0;
};

template <int DIM, int NGL, int MAT, int R, bool DG>
__global__ void __launch_bounds__(256, 4) assemble_ho3_lattice_kernel(Ho3Args T) {
  constexpr int M = NGL - 1, NN = DIM == 3 ? NGL * NGL * NGL : NGL * NGL, GS = DIM == 3 ? 10 : 6;
  constexpr int BRc = MAT == M_LAP ? 1 : DIM;
  constexpr int BCc = MAT == M_K ? DIM : (MAT == M_RW ? (DIM == 3 ? 3 : 1) : 1);
  const int BR = MAT == M_OP ? T.obr : BRc, BC = MAT == M_OP ? T.obc : BCc;   // compile-time constants except for the operators
  const int BB = BR * BC;
  constexpr int BXW = R + 2 * M, BYW = 2 * M + 1, BZW = DIM == 3 ? 2 * M + 1 : 1;   // node box around the run: M nodes on every side
  constexpr int BOX = BXW * BYW * BZW;
  // J^-1, detJ of the elements the run touches: the triple loop makes no global load at all when the table records come from the
  // 1-D factors
  constexpr int GEX = NGL == 3 ? R / 2 + 1 : R + 1, GEL = GEX * (DIM == 3 ? 4 : 2);
  constexpr int NBR = (BOX + 255) / 256, NGR = (GEL * GS + 255) / 256;   // lanes' shares of the Dirichlet bits / the geometry
  static_assert(R + 1 <= 256, "one row offset per lane");
  extern __shared__ double img[];
  __shared__ int srank[2][5], splane[2][5];  // slow axis: sorted position of neighbour plane j, and its inverse
  __shared__ unsigned char nb[BOX];          // Dirichlet bits of the node box (filled in runs that see an imposed DOF only)
  __shared__ double gl[2][GEL * GS];
  __shared__ double t1d[8 * NGL * NGL];
  __shared__ int rcs[R];                     // first block of the run's rows in a compact Krhs (runs with an imposed DOF only)
  const int tid = threadIdx.x;
  const bool tens = T.tabs1d != nullptr;
  if (tens && tid < 8 * NGL * NGL) t1d[tid] = T.tabs1d[tid];
  const int NX = T.NX;
  const int NYL = DIM == 3 ? T.NY : T.npl;   // extent of the y axis
  // lines of one launch are of one class: the number of elements around a line along y / z is the launch's
  const int ny_e = (NGL == 2 || !T.par_y) ? 2 : 1, nz_e = DIM == 3 ? ((NGL == 2 || !T.par_z) ? 2 : 1) : 1;
  const int nyz = ny_e * nz_e, sh = nyz == 4 ? 2 : (nyz == 2 ? 1 : 0);

  auto locate = [&](int w) {
    Ho3Run q;
    const int run = w % T.nruns;
    int bid = w / T.nruns, so;
    q.x0 = run * R;
    q.nrows = min(R, NX - q.x0);
    q.cz = 0;
    if (DIM == 3) {
      const int iy = bid % T.nly;
      so = T.so0 + T.step * (bid / T.nly);
      q.cy = NGL == 3 ? 2 * iy + T.par_y : iy;
      q.cz = T.p_own0 + so;
    } else {
      so = T.so0 + T.step * bid;
      q.cy = T.p_own0 + so;
    }
    q.zlo = 0;
    q.n_z = 1;
    axis_range<NGL>(q.cy, NYL, q.ylo, q.n_y);
    if (DIM == 3) axis_range<NGL>(q.cz, T.npl, q.zlo, q.n_z);
    q.row0 = DIM == 3 ? ((int64_t)so * T.NY + q.cy) * NX + q.x0 : (int64_t)so * NX + q.x0;
    return q;
  };
  // first node id of plane (x-line in 2-D) j: in closed form for the numbering of a rank's slab (owned planes, ghost planes below,
  // ghost planes above; verified on the host), from memory otherwise
  const int PSZ = DIM == 3 ? NX * T.NY : NX;
  auto pbase = [&](int j) -> int {
    if (!T.pstd) return T.P[j];
    const int p0 = T.p_own0, no = T.n_own;
    return (j < p0 ? no + j : (j >= p0 + no ? j : j - p0)) * PSZ;
  };
  // requests (registers) ...
  // (the offset of the line's first row comes through the scalar cache: the vector-memory path is busy with the stores of the runs
  // before; the offsets of the other rows follow in closed form: x_prefix)
  typedef const __attribute__((address_space(4))) int32_t* scalar_i32;
  typedef const __attribute__((address_space(4))) uint32_t* scalar_u32;
  int f_rpl = 0, f_flag = 0, f_p = 0;   // f_p: lane j < n_s of the first wave holds the first id of neighbour plane j
  double f_gl[NGR];
  auto fetch = [&](const Ho3Run& q) {
    const int slo = DIM == 3 ? q.zlo : q.ylo, n_s = DIM == 3 ? q.n_z : q.n_y;
    f_rpl = ((scalar_i32)T.rowptr)[q.row0 - q.x0];
    if (tid < 64) f_p = tid < n_s ? pbase(slo + tid) : INT32_MAX;
    // does the run see an imposed DOF?  (one byte of the flag array, through the scalar cache)
    f_flag = 0;
    if (T.runflag && !(T.ablate & 16)) f_flag = (((scalar_u32)T.runflag)[q.row0 >> 2] >> (8 * (int)(q.row0 & 3))) & 0xff;
#pragma unroll
    for (int u = 0; u < NGR; ++u) {
      const int i = tid + 256 * u;
      f_gl[u] = 0.0;
      if (i < GEX * nyz * GS && !(T.ablate & 1)) {
        const int ge = i / GS, w = i - ge * GS;
        const int gx = ge % GEX, gyz = ge / GEX, ys = gyz % ny_e, zs = gyz / ny_e;
        const int ex = (NGL == 3 ? (q.x0 >> 1) - 1 : q.x0 - 1) + gx;
        const int ey = NGL == 2 ? q.cy - 1 + ys : ((q.cy & 1) ? (q.cy >> 1) : (q.cy >> 1) - 1 + ys);
        const int ez = DIM == 3 ? (NGL == 2 ? q.cz - 1 + zs : ((q.cz & 1) ? (q.cz >> 1) : (q.cz >> 1) - 1 + zs)) : 0;
        if (ex >= 0 && ex < T.EX && ey >= 0 && ey < T.EY && (DIM == 2 || (ez >= 0 && ez < T.EZ)))
          f_gl[u] = T.geom[((T.ablate & 32) ? 0 : (ex + (int64_t)T.EX * (ey + (int64_t)T.EY * ez)) * GS) + w];
      }
    }
  };
  // ... and their place in LDS set `set`
  auto stash = [&](int set, const Ho3Run& q) {
    const int slo = DIM == 3 ? q.zlo : q.ylo, n_s = DIM == 3 ? q.n_z : q.n_y;
    if (tid < 64) {   // (the whole first wave: readlane needs no particular lane alive)
      int rk = 0;
#pragma unroll
      for (int j = 0; j < 5; ++j) rk += __builtin_amdgcn_readlane(f_p, j) < f_p;
      if (tid < n_s) {
        srank[set][tid] = rk;
        splane[set][rk] = slo + tid;
      }
    }
#pragma unroll
    for (int u = 0; u < NGR; ++u)
      if (tid + 256 * u < GEX * nyz * GS) gl[set][tid + 256 * u] = f_gl[u];
  };

  int w = blockIdx.x;
  if (w >= T.nwork) return;
  Ho3Run q = locate(w);
  fetch(q);
  stash(0, q);
  int flag_cur = f_flag, cur = 0, rpl_cur = f_rpl;
  // a lane keeps ONE column node b of the element (its lattice offsets are loop invariants); the 256 / NN groups of NN lanes walk the
  // (row, element) slots of the run
  constexpr int NG = 256 / NN;
  const int grp = tid / NN, b = tid - grp * NN;
  const int lbx = b % NGL, lby = (b / NGL) % NGL, lbz = b / (NGL * NGL);
  const int EYL = T.EY;
  for (;;) {
    for (int i = tid; i < T.img_len; i += 256) img[i] = 0.0;
    const bool routed = flag_cur != 0;
    // a run that sees an imposed DOF: the Dirichlet bits of its node box and the places of its rows in a compact Krhs are asked for
    // now and used after the triple loop (which needs neither)
    int r_nb[NBR], r_rc = -1;
    const bool compact = MAT != M_RW && T.Arhs && T.rcrow;
    if (routed) {
#pragma unroll
      for (int u = 0; u < NBR; ++u) {
        const int i = tid + 256 * u;
        r_nb[u] = 0;
        if (i < BOX) {
          const int bx = i % BXW, by = (i / BXW) % BYW, bz = i / (BXW * BYW);
          const int x = q.x0 - M + bx, y = q.cy - M + by, z = q.cz - M + bz;
          if (x >= 0 && x < NX && y >= 0 && y < NYL && (DIM == 2 || (z >= 0 && z < T.npl))) {
            const int64_t pz = pbase(DIM == 3 ? z : y);
            r_nb[u] = T.nbits[DIM == 3 ? pz + (int64_t)y * NX + x : pz + x];
          }
        }
      }
      if (compact && tid < q.nrows) r_rc = T.rcrow[q.row0 + tid];
    }
    __syncthreads();   // set `cur` is in place, the image is clear
    const int wn = w + (int)gridDim.x;
    const bool more = wn < T.nwork;
    if (more) fetch(locate(wn));
    const int x0 = q.x0, nrows = q.nrows, cy = q.cy, cz = q.cz, ylo = q.ylo, n_y = q.n_y, zlo = q.zlo;
    // row offsets of the run relative to its first block: (x_prefix(x0 + r) - x_prefix(x0)) n_y n_z
    const int nyzc = n_y * q.n_z, px0 = x_prefix<NGL>(x0, NX);
    const int rp0 = rpl_cur + px0 * nyzc;
    auto rel = [&](int r) { return (x_prefix<NGL>(x0 + r, NX) - px0) * nyzc; };
    const int ex0 = NGL == 3 ? (x0 >> 1) - 1 : x0 - 1;             // first element column the run can touch

    // ---- the (row, element, column node) triples of the run.  ngl 3: rows alternate vertex-like (two elements along x) / mid-node
    //      (one), enumerated per row pair with three x-slots; ngl 2: every row has two elements along x
    const int nslot = (NGL == 3 ? ((nrows + 1) >> 1) * 3 : nrows * 2) * nyz;
    for (int t0 = grp; t0 < nslot && grp < NG && !(T.ablate & 1); t0 += NG) {
      const int yz = t0 & (nyz - 1), t1 = t0 >> sh;
      int cx, ex, la_x;
      if (NGL == 3) {
        const int pr = t1 / 3, xs = t1 - pr * 3;
        cx = x0 + 2 * pr + (xs == 2);
        ex = (x0 >> 1) + pr - (xs == 0);
        la_x = xs == 0 ? 2 : (xs == 1 ? 0 : 1);
      } else {
        const int pr = t1 >> 1, xs = t1 & 1;
        cx = x0 + pr;
        ex = cx - 1 + xs;
        la_x = 1 - xs;
      }
      if (cx >= NX || ex < 0 || ex >= T.EX) continue;
      const int ys = yz & (ny_e - 1), zs = ny_e == 2 ? yz >> 1 : yz;
      int ey, la_y, ez = 0, la_z = 0;
      if (NGL == 2) {
        ey = cy - 1 + ys;
        la_y = 1 - ys;
      } else if (cy & 1) {
        ey = cy >> 1;
        la_y = 1;
      } else {
        ey = (cy >> 1) - 1 + ys;
        la_y = ys ? 0 : 2;
      }
      if (ey < 0 || ey >= EYL) continue;
      if (DIM == 3) {
        if (NGL == 2) {
          ez = cz - 1 + zs;
          la_z = 1 - zs;
        } else if (cz & 1) {
          ez = cz >> 1;
          la_z = 1;
        } else {
          ez = (cz >> 1) - 1 + zs;
          la_z = zs ? 0 : 2;
        }
        if (ez < 0 || ez >= T.EZ) continue;
      }
      const double* __restrict__ ge = gl[cur] + ((zs * ny_e + ys) * GEX + (ex - ex0)) * GS;
      const int a = (la_z * NGL + la_y) * NGL + la_x;
      int xlo, n_x;
      axis_range<NGL>(cx, NX, xlo, n_x);
      const int kx = M * ex + lbx - xlo, ky = M * ey + lby - ylo;
      int k;
      if (DIM == 3)
        k = (srank[cur][M * ez + lbz - zlo] * n_y + ky) * n_x + kx;
      else
        k = srank[cur][ky] * n_x + kx;
      const int r = cx - x0;
      const int base = rel(r) * BB, len = n_x * nyzc;
      double v[3][3];
      if (DG) {   // axis-aligned boxes (and 1-D factors): the block straight from them
        const int la[3] = {la_x, la_y, la_z}, lb[3] = {lbx, lby, lbz};
        ho3_block_diag<DIM, NGL, MAT>(ge, t1d, la, lb, T.alpha_d, T.alpha_w, v);
      } else if (tens) {   // the record from the 1-D factors in LDS: the loop makes no global load
        double rec[TabRec<DIM>::TS];
        const int la[3] = {la_x, la_y, la_z}, lb[3] = {lbx, lby, lbz};
        ho3_record_1d<DIM, NGL, MAT>(t1d, la, lb, rec);
        ho3_block<DIM, MAT>(ge, rec, T.alpha_d, T.alpha_w, v);
      } else {
        ho3_block<DIM, MAT>(ge, T.tabs + (a * NN + b) * TabRec<DIM>::TS, T.alpha_d, T.alpha_w, v);
      }
      if (T.ablate & 4) {
        if (v[0][0] == 1.2345e-300) img[0] = v[1][1] + v[2][2] + v[0][1] + v[1][0] + v[0][2] + v[2][0] + v[1][2] + v[2][1];
        continue;
      }
      if (MAT == M_OP) {
        for (int t = 0; t < T.nterms; ++t) {
          const int d = T.t_der[t];
          const double g = d == 0 ? v[0][0] : (d == 1 ? v[0][1] : v[0][2]);
          atomicAdd(&img[base + (T.t_row[t] * len + k) * BC + T.t_col[t]], T.t_coef[t] * g);
        }
      } else {
#pragma unroll
        for (int p = 0; p < BRc; ++p)
#pragma unroll
          for (int q2 = 0; q2 < BCc; ++q2) atomicAdd(&img[base + (p * len + k) * BCc + q2], v[p][q2]);
      }
    }
    __syncthreads();
    if (more) {   // (requested a whole triple loop ago; the run's position is recomputed rather than kept in registers)
      int w2 = wn;
      asm volatile("" : "+s"(w2));
      stash(cur ^ 1, locate(w2));
    }

    // ---- the piece of the value array(s) this run owns
    const int64_t gbase = (int64_t)rp0 * BB;
    double* __restrict__ outA = T.A;
    double* __restrict__ outR = T.Arhs;
    if (routed && !(T.ablate & 2)) {
      // Dirichlet elimination of a run with an imposed DOF in its node box, applied IN the image one (row, column node) pair at a
      // time -- most pairs have neither an imposed row nor an imposed column and are left alone; the Krhs entries go straight to
      // memory (all of a stored row's entries, zeros included: a compact matrix is written in full by every assembly).  The image
      // then leaves through the same coalesced copy as every other run.  (Never the operators: they carry no elimination,
      // mat_generator.py:157-170.)
#pragma unroll
      for (int u = 0; u < NBR; ++u)
        if (tid + 256 * u < BOX) nb[tid + 256 * u] = (unsigned char)r_nb[u];
      if (compact && tid < nrows) rcs[tid] = r_rc;
      __syncthreads();
      const int npairs = rel(nrows);
      for (int pr = tid; pr < npairs; pr += 256) {
        int r = 0;
        while (r + 1 < nrows && pr >= rel(r + 1)) ++r;
        const int k = pr - rel(r);
        const int cx = x0 + r;
        int xlo, n_x;
        axis_range<NGL>(cx, NX, xlo, n_x);
        const int base = rel(r) * BB, len = n_x * nyzc;
        int dx, dy, dz = 0;
        if (DIM == 3) {
          const int nxy = n_x * n_y;
          const int kz = small_div(k, nxy), r2 = k - kz * nxy;
          const int ky = small_div(r2, n_x);
          dx = xlo + (r2 - ky * n_x) - cx;
          dy = ylo + ky - cy;
          dz = splane[cur][kz] - cz;
        } else {
          const int ks = small_div(k, n_x);
          dx = xlo + (k - ks * n_x) - cx;
          dy = splane[cur][ks] - cy;
        }
        const int rowbits = nb[((BZW >> 1) * BYW + M) * BXW + r + M];
        const int colbits = MAT == M_RW ? 0 : nb[((dz + (BZW >> 1)) * BYW + dy + M) * BXW + r + M + dx];
        // Arhs: same place as in A for a matrix with the graph's pattern, the row's own start in a compact one (-1: row not stored)
        int64_t rbase = -1;
        if (MAT != M_RW && outR) rbase = compact ? (rcs[r] >= 0 ? (int64_t)rcs[r] * BB : -1) : gbase + base;
        if (!(rowbits | colbits) && rbase < 0) continue;
        const bool diag = dx == 0 && dy == 0 && dz == 0;
#pragma unroll
        for (int p = 0; p < BRc; ++p)
#pragma unroll
          for (int q2 = 0; q2 < BCc; ++q2) {
            const int idx = (p * len + k) * BCc + q2;
            const double v = img[base + idx];
            double va, vr;
            if ((rowbits >> p) & 1) {   // imposed row: unit diagonal in K and Krhs (mat_generator.py:113-118), nothing in Rw
              va = vr = (MAT != M_RW && diag && q2 == p) ? 1.0 : 0.0;
            } else if ((colbits >> q2) & 1) {
              va = 0.0;                  // imposed column of a free row: -K_e[free, bc] goes to Krhs (base_problem.py:531-533)
              vr = -v;
            } else {
              va = v;
              vr = 0.0;
            }
            if (rowbits | colbits) img[base + idx] = va;
            if (rbase >= 0) outR[rbase + idx] = vr;
          }
      }
      __syncthreads();
    }
    if (!(T.ablate & 2)) {
      const int total = rel(nrows) * BB;
      const bool zr = !routed && outR && !T.rhs_clean && !T.rcrow;   // (a compact matrix stores no row of a run without imposed DOFs)
      if (T.ablate & 8) {   // A/B: one double per lane and store
        for (int i = tid; i < total; i += 256) {
          outA[gbase + i] = img[i];
          if (zr) outR[gbase + i] = 0.0;
        }
      } else {
        // two doubles per lane and store (1 KiB per wave instruction; the piece starts on any double: 8-byte aligned pairs)
        typedef double __attribute__((ext_vector_type(2), aligned(8))) d2u;
        for (int i = 2 * tid; i + 1 < total; i += 512) {
          d2u v2;
          v2.x = img[i];
          v2.y = img[i + 1];
          *reinterpret_cast<d2u*>(outA + gbase + i) = v2;
          if (zr) {
            d2u z2;
            z2.x = z2.y = 0.0;
            *reinterpret_cast<d2u*>(outR + gbase + i) = z2;
          }
        }
        if ((total & 1) && tid == 0) {
          outA[gbase + total - 1] = img[total - 1];
          if (zr) outR[gbase + total - 1] = 0.0;
        }
      }
    }
    if (!more) break;
    __syncthreads();   // the image and set `cur` have been read: the next run may clear / replace them
    w = wn;
    asm volatile("" : "+s"(w));
    q = locate(w);
    cur ^= 1;
    flag_cur = f_flag;
    rpl_cur = f_rpl;
  }
}

// Which runs see an imposed DOF at all?  One byte per owned node that starts a run of R rows: the OR of the Dirichlet bits over the
// node box around the run (M nodes on every side).  Once per Dirichlet set and run length; the assembly reads it through the scalar
// cache and touches the bits themselves only in the runs that need the elimination.
__global__ void ho3_runflag_kernel(Ho3Args T, int dim, int M, int R, int64_t n_owned, uint8_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_owned) return;
  const int NX = T.NX, NYL = dim == 3 ? T.NY : T.npl;
  const int x0 = (int)(i % NX);
  if (x0 % R) return;
  int cy, cz = 0;
  if (dim == 3) {
    cy = (int)((i / NX) % T.NY);
    cz = T.p_own0 + (int)(i / ((int64_t)NX * T.NY));
  } else {
    cy = T.p_own0 + (int)(i / NX);
  }
  int any = 0;
  for (int z = (dim == 3 ? cz - M : 0); z <= (dim == 3 ? cz + M : 0); ++z) {
    if (dim == 3 && (z < 0 || z >= T.npl)) continue;
    for (int y = cy - M; y <= cy + M; ++y) {
      if (y < 0 || y >= NYL) continue;
      const int64_t base = dim == 3 ? (int64_t)T.P[z] + (int64_t)y * NX : (int64_t)T.P[y];
      for (int x = max(0, x0 - M); x <= min(NX - 1, x0 + R - 1 + M); ++x) any |= T.nbits[base + x];
    }
  }
  flag[i] = (uint8_t)any;
}

// ---- symbolic phase in closed form: row lengths -> scan -> sorted columns
__device__ __forceinline__ void ho3_row_coords(const Ho3Args& T, int dim, int64_t i, int& cx, int& cy, int& cz) {
  cx = (int)(i % T.NX);
  if (dim == 3) {
    cy = (int)((i / T.NX) % T.NY);
    cz = T.p_own0 + (int)(i / ((int64_t)T.NX * T.NY));
  } else {
    cy = T.p_own0 + (int)(i / T.NX);
    cz = 0;
  }
}

__device__ __forceinline__ void axis_range_rt(int ngl, int c, int N, int& lo, int& n) {
  if (ngl == 2)
    axis_range<2>(c, N, lo, n);
  else
    axis_range<3>(c, N, lo, n);
}

__global__ void ho3_rowlen_kernel(Ho3Args T, int dim, int ngl, int64_t n_rows, int32_t* __restrict__ len) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_rows) return;
  if (i == n_rows) {
    len[i] = 0;
    return;
  }
  int cx, cy, cz, lo, n_x, n_y, n_z = 1;
  ho3_row_coords(T, dim, i, cx, cy, cz);
  axis_range_rt(ngl, cx, T.NX, lo, n_x);
  axis_range_rt(ngl, cy, dim == 3 ? T.NY : T.npl, lo, n_y);
  if (dim == 3) axis_range_rt(ngl, cz, T.npl, lo, n_z);
  len[i] = n_x * n_y * n_z;
}

__global__ void ho3_columns_kernel(Ho3Args T, int dim, int ngl, int64_t n_rows, const int32_t* __restrict__ rowptr,
                                   int32_t* __restrict__ colidx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  int cx, cy, cz, xlo, ylo, zlo = 0, n_x, n_y, n_z = 1;
  ho3_row_coords(T, dim, i, cx, cy, cz);
  axis_range_rt(ngl, cx, T.NX, xlo, n_x);
  axis_range_rt(ngl, cy, dim == 3 ? T.NY : T.npl, ylo, n_y);
  if (dim == 3) axis_range_rt(ngl, cz, T.npl, zlo, n_z);
  const int slo = dim == 3 ? zlo : ylo, n_s = dim == 3 ? n_z : n_y;
  int k = rowptr[i];
  // planes (x-lines in 2-D) in ascending order of their first node id: selection over <= 5 candidates
  int32_t prev = -1;
  for (int t = 0; t < n_s; ++t) {
    int32_t best = INT32_MAX;
    for (int j = 0; j < n_s; ++j) {
      const int32_t pj = T.P[slo + j];
      if (pj > prev && pj < best) best = pj;
    }
    prev = best;
    if (dim == 3) {
      for (int y = ylo; y < ylo + n_y; ++y)
        for (int x = xlo; x < xlo + n_x; ++x) colidx[k++] = best + y * T.NX + x;
    } else {
      for (int x = xlo; x < xlo + n_x; ++x) colidx[k++] = best + x;
    }
  }
}

void fill_lattice_args(const pyn_ctx* c, Ho3Args& T) {
  const Ho3Lattice& L = c->ho3;
  T.EX = L.EX;
  T.EY = L.EY;
  T.EZ = L.EZ;
  T.NX = L.NX;
  T.NY = L.NY;
  T.npl = L.npl;
  T.p_own0 = L.p_own0;
  T.n_own = L.n_own;
  T.P = L.d_P;
  T.rowptr = c->d_rowptr;
  T.nbits = nullptr;
  T.runflag = nullptr;
  T.geom = L.d_geom;
  T.tabs = c->d_ho3_tabs;
  T.tabs1d = c->ho3_tens_ok ? c->d_ho3_t1d : nullptr;
  T.alpha_d = T.alpha_w = 0.0;
  T.A = T.Arhs = nullptr;
  T.rhs_clean = 0;
  T.rcrow = nullptr;
  T.par_y = T.par_z = 0;
  T.nruns = T.nly = T.nwork = 0;
  T.so0 = 0;
  T.img_len = 0;
  T.step = L.ngl == 3 ? 2 : 1;
  T.diag = L.diag == 1 && !getenv("PYNAMA_HO3_NO_DIAG");
  {
    const int64_t PS = L.dim == 3 ? (int64_t)L.NX * L.NY : L.NX;
    bool std_p = !getenv("PYNAMA_HO3_NO_PSTD") && PS * L.npl < (int64_t)INT32_MAX;
    for (int j = 0; j < L.npl && std_p; ++j) {
      const int64_t want = (j < L.p_own0 ? L.n_own + j : (j >= L.p_own0 + L.n_own ? j : j - L.p_own0)) * PS;
      std_p = L.P[j] == want;
    }
    T.pstd = std_p;
  }
  {
    const char* ab = getenv("PYNAMA_HO3_ABLATE");
    T.ablate = ab ? atoi(ab) : 0;
  }
  T.obr = T.obc = 1;
  T.nterms = 0;
}

template <int DIM, int NGL, int MAT, int R, bool DG>
int launch_ho3_g(pyn_ctx* c, Ho3Args T) {
  const int BR = MAT == M_OP ? T.obr : (MAT == M_LAP ? 1 : DIM);
  const int BC = MAT == M_OP ? T.obc : (MAT == M_K ? DIM : (MAT == M_RW ? (DIM == 3 ? 3 : 1) : 1));
  Ho3Lattice& L = c->ho3;
  T.nruns = (L.NX + R - 1) / R;
  static bool attr_done = false;
  if (!attr_done) {
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_ho3_lattice_kernel<DIM, NGL, MAT, R, DG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    attr_done = true;
  }
  T.runflag = nullptr;
  if (T.nbits) {   // which runs see an imposed DOF: once per Dirichlet set and run length
    if (L.runflag_stamp != c->bc_stamp || L.runflag_R != R) {
      if (!L.d_runflag) PYN_HIP(hipMalloc((void**)&L.d_runflag, (size_t)c->n_owned + 8));
      ho3_runflag_kernel<<<(int)((c->n_owned + 255) / 256), 256, 0, c->stream>>>(T, DIM, NGL - 1, R, c->n_owned, L.d_runflag);
      PYN_HIP(hipGetLastError());
      L.runflag_stamp = c->bc_stamp;
      L.runflag_R = R;
    }
    T.runflag = L.d_runflag;
  }
  // ngl 3: one launch per class of x-lines (parity of y, z), its LDS sized for that class; ngl 2: every line is of one class
  const int ncls = NGL == 3 ? 2 : 1;
  for (int pz = 0; pz < (DIM == 3 ? ncls : 1); ++pz)
    for (int py = 0; py < ncls; ++py) {
      T.par_y = py;
      T.par_z = pz;
      int nslow;
      if (NGL == 3) {
        const int ps = DIM == 3 ? pz : py;                      // parity along the slow axis
        T.so0 = ((L.p_own0 & 1) == ps) ? 0 : 1;
        nslow = T.so0 < L.n_own ? (L.n_own - T.so0 + 1) / 2 : 0;
        T.nly = DIM == 3 ? (py == 0 ? L.EY + 1 : L.EY) : 1;
      } else {
        T.so0 = 0;
        nslow = L.n_own;
        T.nly = DIM == 3 ? L.NY : 1;
      }
      const int64_t grid = (int64_t)T.nruns * T.nly * nslow;
      if (grid == 0) continue;
      PYN_CHECK(grid < (int64_t)INT32_MAX, "lattice row-run assembly: %lld workgroups", (long long)grid);
      if (NGL == 3) {
        const int n_y = py ? 3 : 5, n_z = DIM == 3 ? (pz ? 3 : 5) : 1;
        T.img_len = ((R + 1) / 2) * (5 + 3) * n_y * n_z * BR * BC;
      } else {
        T.img_len = R * (DIM == 3 ? 27 : 9) * BR * BC;
      }
      const size_t lds = (size_t)T.img_len * sizeof(double);
      PYN_CHECK(lds <= 128 * 1024, "lattice row-run assembly: %zu B of LDS per run", lds);
      // as many workgroups as the device holds at once; each walks its share of the runs (PYNAMA_HO3_WGS_PER_CU: fewer / more)
      static size_t occ_lds[4] = {0, 0, 0, 0};
      static int occ_wgs[4] = {0, 0, 0, 0};
      const int cls = 2 * pz + py;
      if (occ_lds[cls] != lds || !occ_wgs[cls]) {
        PYN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_wgs[cls], assemble_ho3_lattice_kernel<DIM, NGL, MAT, R, DG>, 256, lds));
        occ_lds[cls] = lds;
      }
      int per_cu = std::max(1, occ_wgs[cls]);
      if (const char* e = getenv("PYNAMA_HO3_WGS_PER_CU")) per_cu = std::max(1, atoi(e));
      T.nwork = (int)grid;
      int64_t launch = std::min<int64_t>(grid, (int64_t)per_cu * 256);   // 256 CUs (gfx950)
      if (const char* e = getenv("PYNAMA_HO3_GRID")) launch = std::min<int64_t>(grid, std::max(1, atoi(e)));   // tests: many runs per workgroup
      assemble_ho3_lattice_kernel<DIM, NGL, MAT, R, DG><<<(int)launch, 256, lds, c->stream>>>(T);
    }
  PYN_HIP(hipGetLastError());
  return PYN_OK;
}

// axis-aligned boxes with verified 1-D factors take the kernels that build the blocks straight from them
template <int DIM, int NGL, int MAT, int R>
int launch_ho3(pyn_ctx* c, const Ho3Args& T) {
  if (T.diag && T.tabs1d) return launch_ho3_g<DIM, NGL, MAT, R, true>(c, T);
  return launch_ho3_g<DIM, NGL, MAT, R, false>(c, T);
}

// rows per run: tuned on 1024^2 / 64^3 (ngl 3); PYNAMA_HO3_RUN overrides (tests walk every length)
template <int DIM, int MAT>
int launch_ho3_r(pyn_ctx* c, const Ho3Args& T) {
  const char* e = getenv("PYNAMA_HO3_RUN");
  const int r = e ? atoi(e) : 0;
  if (c->ho3.ngl == 2) {
    if (DIM == 3) return launch_ho3<3, 2, MAT, 8>(c, T);
    return launch_ho3<2, 2, MAT, 32>(c, T);
  }
  if (DIM == 3) {
    if (MAT == M_OP) return launch_ho3<3, 3, MAT, 2>(c, T);      // up to 6 x 3 values per graph edge: two rows per run fill the LDS
    if (r == 2) return launch_ho3<3, 3, MAT, 2>(c, T);
    if (r == 8) return launch_ho3<3, 3, MAT, 8>(c, T);
    return launch_ho3<3, 3, MAT, 4>(c, T);
  }
  if (MAT == M_OP) return launch_ho3<2, 3, MAT, 16>(c, T);
  if (r == 16) return launch_ho3<2, 3, MAT, 16>(c, T);
  if (r == 64) return launch_ho3<2, 3, MAT, 64>(c, T);
  return launch_ho3<2, 3, MAT, 32>(c, T);
}

// geometry pre-pass (+ the once-per-mesh check that every cell is a parallelogram / parallelepiped); *ok = the closed forms apply
int ho3_prepare(pyn_ctx* c, bool* ok) {
  *ok = false;
  Ho3Lattice& L = c->ho3;
  hipStream_t s = c->stream;
  const int gs = L.dim == 3 ? 10 : 6;
  if (!L.d_geom) PYN_HIP(hipMalloc((void**)&L.d_geom, (size_t)c->n_elem * gs * sizeof(double)));
  const int ge = (int)((c->n_elem + 255) / 256);
  DevTmp flag;
  int* d_flag = nullptr;
  if (L.affine < 0) {   // once per mesh
    PYN_HIP(flag.alloc(2 * sizeof(int)));   // [0] some cell is not affine, [1] some J is not diagonal
    PYN_HIP(hipMemsetAsync(flag.p, 0, 2 * sizeof(int), s));
    d_flag = flag.as<int>();
  }
  // J^-1, detJ of every element (part of the numeric phase: runs inside the timed region of every assembly)
  if (L.dim == 3)
    ho3_geom_kernel<3><<<ge, 256, 0, s>>>(c->d_conn, c->d_xyz, c->n_elem, c->nn, c->quad[0].HrsCoo, L.d_geom, d_flag);
  else
    ho3_geom_kernel<2><<<ge, 256, 0, s>>>(c->d_conn, c->d_xyz, c->n_elem, c->nn, c->quad[0].HrsCoo, L.d_geom, d_flag);
  PYN_HIP(hipGetLastError());
  if (d_flag) {
    int h[2] = {1, 1};
    PYN_HIP(hipMemcpyAsync(h, d_flag, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    PYN_HIP(hipStreamSynchronize(s));
    L.affine = h[0] ? 0 : 1;
    L.diag = (h[0] || h[1]) ? 0 : 1;
  }
  *ok = L.affine == 1 && !getenv("PYNAMA_NO_HO3_LATTICE");
  return PYN_OK;
}

}  // namespace

void pyn_ho3_release(pyn_ctx* c) {
  Ho3Lattice& L = c->ho3;
  (void)hipFree(L.d_P);
  (void)hipFree(L.d_geom);
  (void)hipFree(L.d_nbits);
  (void)hipFree(L.d_runflag);
  L = Ho3Lattice();
}

// Is the connectivity that of a structured mesh of tensor-product cells of order 1 or 2 (ngl 2 / 3: the reference's box mesh,
// src/domain/dmplex.py:8-21, 42-61, or a rank's slab of one)?  Host, once per pyn_mesh_set; every entry of `conn` is checked against
// the closed form the kernels use.
// every element of a structured block against its closed form (one thread per entry of the connectivity)
struct LocTab {
  signed char v[27][3];
};
__global__ void ho3_conn_verify_kernel(const int32_t* __restrict__ conn, const int32_t* __restrict__ P, LocTab loc, int dim, int nn, int m,
                                       int64_t ne, int EX, int EY, int NX, int64_t per_layer, int* __restrict__ bad) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ne * nn) return;
  const int64_t e = t / nn;
  const int a = (int)(t - e * nn);
  const int ex = (int)(e % EX), ey = dim == 3 ? (int)((e / EX) % EY) : 0;
  const int64_t el = e / per_layer;
  int64_t id;
  if (dim == 3)
    id = (int64_t)P[m * el + loc.v[a][2]] + (int64_t)(m * ey + loc.v[a][1]) * NX + m * ex + loc.v[a][0];
  else
    id = (int64_t)P[m * el + loc.v[a][1]] + m * ex + loc.v[a][0];
  if (conn[t] != id) atomicAdd(bad, 1);
}

// `at(i)`: entry i of the local connectivity as the host sees it; the shape guessed from O(element rows + layers) entries is checked
// against all of c->d_conn on the device
int pyn_ho3_detect(pyn_ctx* c, const ConnAt& at) {
  pyn_ho3_release(c);
  const int dim = c->dim, nn = c->nn;
  const int ngl = (nn == 9 || nn == 27) ? 3 : 2;
  if (!((dim == 2 && (nn == 9 || nn == 4)) || (dim == 3 && (nn == 27 || nn == 8))) || c->n_elem < 1 || getenv("PYNAMA_NO_HO3")) return PYN_OK;
  const int m = ngl - 1;
  int a_of[27];
  for (int a = 0; a < nn; ++a) a_of[tens_of(dim, ngl, a)] = a;
  const int a0 = a_of[0];
  const int64_t ne = c->n_elem;
  const int32_t c0 = at(a0);
  int64_t EX = 1;
  while (EX < ne && at(EX * nn + a0) == c0 + m * EX) ++EX;
  if (ne % EX) return PYN_OK;
  const int64_t NX = m * EX + 1;
  int64_t EY, EZ = 0, NY = 0, PS, EL;   // EL: element layers along the slow axis
  if (dim == 3) {
    EY = 1;
    while (EY * EX < ne && at(EY * EX * nn + a0) == c0 + m * EY * NX) ++EY;
    if ((ne / EX) % EY) return PYN_OK;
    EZ = ne / (EX * EY);
    NY = m * EY + 1;
    PS = NX * NY;
    EL = EZ;
  } else {
    EY = ne / EX;
    PS = NX;
    EL = EY;
  }
  const int64_t npl = m * EL + 1;
  if (PS * npl != c->n_node || PS > INT32_MAX / 4) return PYN_OK;
  std::vector<int32_t> P((size_t)npl, -1);
  const int64_t per_layer = ne / EL;
  const int stride_s = dim == 3 ? ngl * ngl : ngl;     // tensor stride of the slow axis
  for (int64_t l = 0; l < EL; ++l)
    for (int j = 0; j < ngl; ++j) {
      const int32_t base = at(l * per_layer * nn + a_of[j * stride_s]);
      if (P[m * l + j] >= 0 && P[m * l + j] != base) return PYN_OK;
      P[m * l + j] = base;
    }
  std::vector<int32_t> sorted(P);
  std::sort(sorted.begin(), sorted.end());
  for (int64_t j = 0; j < npl; ++j)
    if (sorted[j] != j * PS) return PYN_OK;
  if (c->n_owned % PS) return PYN_OK;
  const int n_own = (int)(c->n_owned / PS);
  int p0 = -1;
  for (int64_t j = 0; j < npl; ++j)
    if (P[j] == 0) p0 = (int)j;
  if (p0 < 0 || p0 + n_own > npl || n_own < 1) return PYN_OK;
  for (int j = 0; j < n_own; ++j)
    if (P[p0 + j] != (int64_t)j * PS) return PYN_OK;
  Ho3Lattice& L = c->ho3;
  PYN_HIP(hipMalloc((void**)&L.d_P, npl * sizeof(int32_t)));
  PYN_HIP(hipMemcpy(L.d_P, P.data(), npl * sizeof(int32_t), hipMemcpyHostToDevice));
  {   // every element against the guessed shape
    int* d_bad = nullptr;
    int bad = 0;
    PYN_HIP(hipMalloc((void**)&d_bad, sizeof(int)));
    PYN_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), c->stream));
    const unsigned grid = (unsigned)((ne * nn + 255) / 256);
    LocTab lt;
    for (int a = 0; a < nn; ++a)
      for (int d = 0; d < dim; ++d) lt.v[a][d] = (signed char)loc_of(dim, ngl, a, d);
    ho3_conn_verify_kernel<<<grid, 256, 0, c->stream>>>(c->d_conn, L.d_P, lt, dim, nn, m, ne, (int)EX, (int)EY, (int)NX, per_layer, d_bad);
    PYN_HIP(hipGetLastError());
    PYN_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PYN_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(d_bad);
    if (bad) {
      (void)hipFree(L.d_P);
      L.d_P = nullptr;
      return PYN_OK;
    }
  }
  L.P = P;
  L.dim = dim;
  L.ngl = ngl;
  L.EX = (int)EX;
  L.EY = (int)EY;
  L.EZ = (int)EZ;
  L.NX = (int)NX;
  L.NY = (int)NY;
  L.npl = (int)npl;
  L.p_own0 = p0;
  L.n_own = n_own;
  L.valid = true;
  return PYN_OK;
}

// Reference matrices of the element from one uploaded rule (pyn_elem_tables_set): the part of every node pair's record (TabRec) that
// rule supplies -- Tf / Uf (full), Tr / Ur (reduced), Un (nodal) -- in tensor order
int pyn_ho3_tables(pyn_ctx* c, int which, int ngp, const double* w, const double* H, const double* Hrs) {
  const int dim = c->dim, nn = c->nn;
  if (which < 0 || which > 2) return PYN_OK;
  if (!((dim == 2 && (nn == 9 || nn == 4)) || (dim == 3 && (nn == 27 || nn == 8)))) return PYN_OK;
  const int ngl = (nn == 9 || nn == 27) ? 3 : 2;
  const int dd = dim * dim, n2 = nn * nn;
  const int ts = (2 * dd + 3 * dim + 1) & ~1;        // record per node pair (TabRec)
  const size_t total = (size_t)ts * n2;
  if (c->ho3_tabs_nn != nn) {
    (void)hipFree(c->d_ho3_tabs);
    c->d_ho3_tabs = nullptr;
    c->ho3_tabs_ok[0] = c->ho3_tabs_ok[1] = c->ho3_tabs_ok[2] = false;
    PYN_HIP(hipMalloc((void**)&c->d_ho3_tabs, total * sizeof(double)));
    PYN_HIP(hipMemsetAsync(c->d_ho3_tabs, 0, total * sizeof(double), c->stream));
    c->ho3_tabs_host.assign(total, 0.0);
    c->ho3_tabs_nn = nn;
  }
  std::vector<double>& R = c->ho3_tabs_host;
  for (int a = 0; a < nn; ++a)
    for (int b = 0; b < nn; ++b) {
      const int ta = tens_of(dim, ngl, a), tb = tens_of(dim, ngl, b);
      double* rec = R.data() + (size_t)(ta * nn + tb) * ts;
      double* recT = R.data() + (size_t)(tb * nn + ta) * ts;      // Ur is read transposed
      for (int r = 0; r < dim; ++r) {
        if (which != PYN_Q_NODAL)
          for (int s = 0; s < dim; ++s) {
            double acc = 0.0;
            for (int g = 0; g < ngp; ++g) acc += w[g] * Hrs[((size_t)g * dim + r) * nn + a] * Hrs[((size_t)g * dim + s) * nn + b];
            rec[(which == PYN_Q_FULL ? 0 : dd) + r * dim + s] = acc;
          }
        double acc = 0.0;
        for (int g = 0; g < ngp; ++g) acc += w[g] * H[(size_t)g * nn + a] * Hrs[((size_t)g * dim + r) * nn + b];
        if (which == PYN_Q_FULL) rec[2 * dd + r] = acc;
        if (which == PYN_Q_RED) recT[2 * dd + dim + r] = acc;
        if (which == PYN_Q_NODAL) rec[2 * dd + 2 * dim + r] = acc;
      }
    }
  PYN_HIP(hipMemcpyAsync(c->d_ho3_tabs, R.data(), total * sizeof(double), hipMemcpyHostToDevice, c->stream));
  // 1-D factors of this rule: M[i][j] = sum w h_i h_j, D[i][j] = sum w h'_i h_j, S[i][j] = sum w h'_i h'_j, recovered from the uploaded
  // tensor tables by summing out the other axes (sum_ij M_ij = sum w = 2 on [-1, 1]); the records rebuilt from them are checked below
  {
    const int n1 = ngl * ngl;
    if ((int)c->ho3_t1d_host.size() != 8 * n1) c->ho3_t1d_host.assign((size_t)8 * n1, 0.0);
    double* M1 = c->ho3_t1d_host.data() + (size_t)(which == PYN_Q_NODAL ? 6 : 3 * which) * n1;
    double* D1 = M1 + n1;
    double* S1 = which == PYN_Q_NODAL ? nullptr : M1 + 2 * n1;
    for (int k = 0; k < n1; ++k) {
      M1[k] = D1[k] = 0.0;
      if (S1) S1[k] = 0.0;
    }
    double norm = 1.0;
    for (int d = 1; d < dim; ++d) norm *= 2.0;
    for (int a = 0; a < nn; ++a)
      for (int b = 0; b < nn; ++b) {
        const int i = loc_of(dim, ngl, a, 0), j = loc_of(dim, ngl, b, 0);
        double mass = 0.0, u0 = 0.0, t00 = 0.0;
        for (int g = 0; g < ngp; ++g) {
          mass += w[g] * H[(size_t)g * nn + a] * H[(size_t)g * nn + b];
          u0 += w[g] * Hrs[((size_t)g * dim + 0) * nn + a] * H[(size_t)g * nn + b];                       // h'_a h_b along x
          t00 += w[g] * Hrs[((size_t)g * dim + 0) * nn + a] * Hrs[((size_t)g * dim + 0) * nn + b];
        }
        M1[i * ngl + j] += mass / norm;
        D1[i * ngl + j] += u0 / norm;
        if (S1) S1[i * ngl + j] += t00 / norm;
      }
  }
  c->ho3_tabs_ok[which] = true;
  c->ho3_tens_ok = false;
  if (c->ho3_tabs_ok[0] && c->ho3_tabs_ok[1] && c->ho3_tabs_ok[2] && !getenv("PYNAMA_HO3_DENSE_TABLES")) {
    // every record entry against the product of the 1-D factors
    const int n1 = ngl * ngl;
    const double* t1 = c->ho3_t1d_host.data();
    double worst = 0.0, scale = 0.0;
    for (int ta = 0; ta < nn; ++ta)
      for (int tb = 0; tb < nn; ++tb) {
        int la[3] = {0, 0, 0}, lb[3] = {0, 0, 0};
        for (int d = 0, x = ta, y = tb; d < dim; ++d, x /= ngl, y /= ngl) {
          la[d] = x % ngl;
          lb[d] = y % ngl;
        }
        auto fac = [&](int tab, int d, bool tr) { return t1[tab * n1 + (tr ? lb[d] * ngl + la[d] : la[d] * ngl + lb[d])]; };
        const double* rec = R.data() + (size_t)(ta * nn + tb) * ts;
        for (int rule = 0; rule < 2; ++rule)
          for (int r = 0; r < dim; ++r)
            for (int s2 = 0; s2 < dim; ++s2) {
              double p = 1.0;
              for (int d = 0; d < dim; ++d)
                p *= (d == r && d == s2) ? fac(3 * rule + 2, d, false) : (d == r ? fac(3 * rule + 1, d, false) : (d == s2 ? fac(3 * rule + 1, d, true) : fac(3 * rule, d, false)));
              const double ref = rec[rule * dd + r * dim + s2];
              worst = std::max(worst, std::fabs(p - ref));
              scale = std::max(scale, std::fabs(ref));
            }
        for (int x = 0; x < dim; ++x) {
          double pf = fac(1, x, true), pr = fac(4, x, false), pn = fac(7, x, true);
          for (int d = 0; d < dim; ++d)
            if (d != x) {
              pf *= fac(0, d, false);
              pr *= fac(3, d, false);
              pn *= fac(6, d, false);
            }
          worst = std::max(worst, std::max(std::fabs(pf - rec[2 * dd + x]), std::max(std::fabs(pr - rec[2 * dd + dim + x]), std::fabs(pn - rec[2 * dd + 2 * dim + x]))));
        }
      }
    if (worst <= 1e-13 * scale) {
      if (!c->d_ho3_t1d) PYN_HIP(hipMalloc((void**)&c->d_ho3_t1d, 8 * 9 * sizeof(double)));
      PYN_HIP(hipMemcpyAsync(c->d_ho3_t1d, t1, (size_t)8 * n1 * sizeof(double), hipMemcpyHostToDevice, c->stream));
      c->ho3_tens_ok = true;
    }
  }
  PYN_HIP(hipStreamSynchronize(c->stream));
  return PYN_OK;
}

int pyn_ho3_symbolic(pyn_ctx* c, bool* done) {
  *done = false;
  const Ho3Lattice& L = c->ho3;
  if (!L.valid || getenv("PYNAMA_NO_HO3_SYMBOLIC")) return PYN_OK;
  hipStream_t s = c->stream;
  Ho3Args T;
  fill_lattice_args(c, T);
  const int64_t n = c->n_owned;
  DevTmp tlen, tmp;
  PYN_HIP(tlen.alloc((n + 1) * sizeof(int32_t)));
  (void)hipFree(c->d_rowptr);
  (void)hipFree(c->d_colidx);
  c->d_rowptr = nullptr;
  c->d_colidx = nullptr;
  c->nnzb = 0;
  PYN_HIP(hipMalloc((void**)&c->d_rowptr, (n + 1) * sizeof(int32_t)));
  const int grid = (int)((n + 1 + 255) / 256);
  ho3_rowlen_kernel<<<grid, 256, 0, s>>>(T, L.dim, L.ngl, n, tlen.as<int32_t>());
  // the total must fit the int32 CSR before the scan wraps: the interior count bounds it
  const double per_elem = L.ngl == 3 ? (L.dim == 3 ? 512.0 : 64.0) : (L.dim == 3 ? 27.0 : 9.0);
  const double est = (double)c->n_elem * per_elem;
  PYN_CHECK(est < 2.0e9, "pattern has about %.3g entries (int32 CSR limit)", est);
  size_t tb = 0;
  PYN_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, tlen.as<int32_t>(), c->d_rowptr, (int)(n + 1), s));
  PYN_HIP(tmp.alloc(tb));
  PYN_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, tlen.as<int32_t>(), c->d_rowptr, (int)(n + 1), s));
  int32_t nnz = 0;
  PYN_HIP(hipMemcpyAsync(&nnz, c->d_rowptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PYN_HIP(hipStreamSynchronize(s));
  PYN_CHECK(nnz > 0, "empty pattern");
  PYN_HIP(hipMalloc((void**)&c->d_colidx, (size_t)nnz * sizeof(int32_t)));
  ho3_columns_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(T, L.dim, L.ngl, n, c->d_rowptr, c->d_colidx);
  PYN_HIP(hipGetLastError());
  c->nnzb = nnz;
  *done = true;
  return PYN_OK;
}

// K (+ Krhs), Rw of pyn_assemble_kle and the scalar Laplacian of pyn_assemble_scalar on a structured mesh of parallelograms /
// parallelepipeds: second-order cells (ngl 3) in 2-D and 3-D, first-order quadrilaterals (3-D first-order cells have kernels of their
// own, pyn_assemble_lattice.hip).  *handled stays false when the mesh or the tables do not fit (the caller falls back).
int pyn_assemble_ho3_lattice(pyn_ctx* c, int form, double alpha_d, double alpha_w, double* K, double* Krhs, double* Rw, bool* handled) {
  *handled = false;
  Ho3Lattice& L = c->ho3;
  if (!L.valid || c->ho3_tabs_nn != c->nn || !c->ho3_tabs_ok[0] || c->quad[0].ngp < 1) return PYN_OK;
  if (L.ngl == 2 && L.dim == 3) return PYN_OK;
  if (form == PYN_FORM_KLE && !c->ho3_tabs_ok[1]) return PYN_OK;
  if (form != PYN_FORM_KLE && form != PYN_FORM_LAPLACE) return PYN_OK;
  hipStream_t s = c->stream;
  bool ok = false;
  PYN_TRY(ho3_prepare(c, &ok));
  if (!ok) return PYN_OK;
  Ho3Args T;
  fill_lattice_args(c, T);
  if (c->d_bcmask) {
    if (L.nbits_stamp != c->bc_stamp) {
      if (!L.d_nbits) PYN_HIP(hipMalloc((void**)&L.d_nbits, (size_t)c->n_node));
      ho3_pack_bits_kernel<<<(int)((c->n_node + 255) / 256), 256, 0, s>>>(c->d_bcmask, c->n_node, c->bc_ndof, L.d_nbits);
      L.nbits_stamp = c->bc_stamp;
    }
    T.nbits = L.d_nbits;
  }
  T.alpha_d = alpha_d;
  T.alpha_w = alpha_w;
  if (form == PYN_FORM_LAPLACE) {
    PYN_CHECK(K, "scalar assembly without a target");
    T.A = K;
    T.Arhs = Krhs;
    T.rhs_clean = c->asm_rhs_clean ? 1 : 0;
    T.rcrow = c->asm_rcrow;
    if (L.dim == 3)
      PYN_TRY((launch_ho3_r<3, M_LAP>(c, T)));
    else
      PYN_TRY((launch_ho3_r<2, M_LAP>(c, T)));
    *handled = true;
    return PYN_OK;
  }
  if (K) {
    T.A = K;
    T.Arhs = Krhs;
    T.rhs_clean = c->asm_rhs_clean ? 1 : 0;
    T.rcrow = c->asm_rcrow;
    if (L.dim == 3)
      PYN_TRY((launch_ho3_r<3, M_K>(c, T)));
    else
      PYN_TRY((launch_ho3_r<2, M_K>(c, T)));
  }
  if (Rw) {
    T.A = Rw;
    T.Arhs = nullptr;
    T.rhs_clean = 0;
    T.rcrow = nullptr;
    if (L.dim == 3)
      PYN_TRY((launch_ho3_r<3, M_RW>(c, T)));
    else
      PYN_TRY((launch_ho3_r<2, M_RW>(c, T)));
  }
  *handled = true;
  return PYN_OK;
}

// The first-order operators SrT / DivSrT / Curl (pyn_assemble_operator at the nodal rule) on any structured mesh the row-run kernels
// know: ngl 2 / 3, 2-D / 3-D, parallelograms / parallelepipeds.
int pyn_assemble_ho3_operator(pyn_ctx* c, int rule, int br, int bc, int nterms, const int32_t* terms, const double* coef, double* M,
                              bool* handled) {
  *handled = false;
  Ho3Lattice& L = c->ho3;
  if (!L.valid || rule != PYN_Q_NODAL || c->ho3_tabs_nn != c->nn || !c->ho3_tabs_ok[2] || c->quad[0].ngp < 1 || nterms > MAX_TERMS ||
      getenv("PYNAMA_NO_HO3_OPERATOR"))
    return PYN_OK;
  bool ok = false;
  PYN_TRY(ho3_prepare(c, &ok));
  if (!ok) return PYN_OK;
  Ho3Args T;
  fill_lattice_args(c, T);
  T.A = M;
  T.obr = br;
  T.obc = bc;
  T.nterms = nterms;
  for (int t = 0; t < nterms; ++t) {
    T.t_row[t] = terms[3 * t];
    T.t_col[t] = terms[3 * t + 1];
    T.t_der[t] = terms[3 * t + 2];
    T.t_coef[t] = coef[t];
  }
  if (L.dim == 3)
    PYN_TRY((launch_ho3_r<3, M_OP>(c, T)));
  else
    PYN_TRY((launch_ho3_r<2, M_OP>(c, T)));
  *handled = true;
  return PYN_OK;
}
