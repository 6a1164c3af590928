// Numeric phase (HOT LOOP 1): per-element quadrature + global scatter with Dirichlet elimination.
//
// Reference: Spectral.getElemKLEMatrices (src/elements/spectral.py:89-157) called per cell by
// FreeSlip.buildKLEMats (src/cases/base_problem.py:499-552), then Mat.setIndices2One
// (src/matrices/mat_generator.py:113-118).
//
// Closed forms used by the kernels (derived from the B-matrix products of spectral.py:124-156;
// G = J^-1 Hrs are physical gradients at a point, c = w detJ, eps = Levi-Civita):
//   K [(a,p),(b,q)] = d_pq sum_full c G_a.G_b
//                   + sum_red c ( alpha_d G_pa G_qb + alpha_w ( d_pq G_a.G_b - G_qa G_pb ) )
//   Rw[(a,p),(b,k)] = sum_full c H_a curlsel(p,k;G_b) + sum_red alpha_w c curlsel(k,p;G_a)^T H_b
//   Rd[(a,p), b   ] = -sum_full c H_a G_pb + sum_red alpha_d c G_pa H_b
// where curlsel encodes (curl w)_p = eps_pmk d_m w_k (3D) and (d_y w, -d_x w) (2D).
#include "pyn_internal.h"

namespace {

struct AsmArgs {
  // mesh
  const int32_t* conn;
  const double* xyz;
  int64_t n_elem, n_owned;
  int dim, nn, nc;
  // tables: 0 = full, 1 = reduced, 2 = nodal
  int ngp[3];
  const double* w[3];
  const double* H[3];
  const double* Hrs[3];
  const double* HrsCoo[3];
  // graph
  const int32_t* rowptr;
  const int32_t* colidx;
  // bc
  const uint8_t* bcmask;  // may be null
  // form
  int form;
  double alpha_d, alpha_w;
  // outputs (scatter mode) -- null when skipped
  double *K, *Krhs, *Rw, *Rd;
  // no-slip / free-slip split (NoSlipFreeSlip.buildKLEMats, base_problem.py:329-454): DOF classes in
  // bcmask are 0 free, 1 tangential at a no-slip wall (free in the FS solve), 2 imposed in both solves
  double *Kfs, *Krhsfs, *Rwfs, *Rdfs;
  // compact imposed-column targets (pyn_rhs.hip): first block of every owned node row in Krhs / Krhsfs, -1 = row not stored; null = the
  // matrix has the graph's full pattern
  const int32_t *rcrow, *rcrow_fs;
  // element subset (elements with an imposed node: the ones that feed an imposed-column matrix); null = all elements
  const int32_t* esel;
  // dense mode
  const double* corners;  // single element
  double *out0, *out1, *out2;
  // first-order operator form: M[(a,p),(b,q)] = sum_g c_g H_g[a] sum_t [row_t=p, col_t=q] coef_t G_g[der_t][b]
  int op_rule, op_br, op_bc, op_nterms;
  const int32_t* op_terms;  // [nterms][3] = (row comp, col comp, derivative axis)
  const double* op_coef;    // [nterms]
  // scratch for high order
  double* gscratch;
  int64_t gscratch_stride;  // doubles per block
  int ho_chunk;             // > 0: high-order KLE path, Gauss points staged through LDS `ho_chunk` at a time
};

__device__ inline int find_slot(const int32_t* __restrict__ colidx, int lo, int len, int col) {
  // lower_bound over a sorted row; the column is guaranteed to be present
  int l = 0, h = len;
  while (l < h) {
    int m = (l + h) >> 1;
    if (colidx[lo + m] < col)
      l = m + 1;
    else
      h = m;
  }
  return l;
}

__device__ inline double inv_det(const double* J, double* Ji, int dim) {
  if (dim == 2) {
    double det = J[0] * J[3] - J[1] * J[2];
    double r = 1.0 / det;
    Ji[0] = J[3] * r;
    Ji[1] = -J[1] * r;
    Ji[2] = -J[2] * r;
    Ji[3] = J[0] * r;
    return det;
  }
  double c00 = J[4] * J[8] - J[5] * J[7];
  double c01 = J[5] * J[6] - J[3] * J[8];
  double c02 = J[3] * J[7] - J[4] * J[6];
  double det = J[0] * c00 + J[1] * c01 + J[2] * c02;
  double r = 1.0 / det;
  Ji[0] = c00 * r;
  Ji[1] = (J[2] * J[7] - J[1] * J[8]) * r;
  Ji[2] = (J[1] * J[5] - J[2] * J[4]) * r;
  Ji[3] = c01 * r;
  Ji[4] = (J[0] * J[8] - J[2] * J[6]) * r;
  Ji[5] = (J[2] * J[3] - J[0] * J[5]) * r;
  Ji[6] = c02 * r;
  Ji[7] = (J[1] * J[6] - J[0] * J[7]) * r;
  Ji[8] = (J[0] * J[4] - J[1] * J[3]) * r;
  return det;
}

// (curl w)_p = sum_{m,k} cs(p,m,k) d_m w_k ; returns the sign and the derivative index m for (p,k), 0 if none
__device__ inline int curl_term(int dim, int p, int k, int* m) {
  if (dim == 2) {  // w scalar (k = 0): (curl w)_x = d_y w, (curl w)_y = -d_x w
    *m = 1 - p;
    return p == 0 ? 1 : -1;
  }
  if (p == k) return 0;
  *m = 3 - p - k;
  // eps_{p m k}
  return ((m[0] - p + 3) % 3 == 1) ? 1 : -1;
}

// Generic element kernel: one workgroup per element (grid-stride).  All per-point data
// (J^-1, c, G) live in `pt` -- LDS when it fits (PT_LDS), per-block global scratch otherwise.
// v[m] for a runtime m without dynamic register-array indexing (which would put the array in scratch memory)
__device__ __forceinline__ double pick3(const double (&v)[3], int m) { return m == 0 ? v[0] : (m == 1 ? v[1] : v[2]); }

template <int BLOCK, bool PT_LDS, bool DENSE>
__global__ void __launch_bounds__(BLOCK) assemble_generic_kernel(AsmArgs A) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int dim = A.dim, nn = A.nn, nc = A.nc;
  const int dd = dim * dim;
  // which rules this form integrates on
  const bool oper = A.form == PYN_FORM_OPERATOR;
  const int qa = oper ? A.op_rule : (A.form == PYN_FORM_MASS_NODAL) ? 2 : 0;  // primary rule
  const bool kle = A.form == PYN_FORM_KLE;
  const int nga = A.ngp[qa];
  const int ngb = kle ? A.ngp[1] : 0;  // reduced rule only for KLE
  const int ngt = nga + ngb;

  double* Xs = reinterpret_cast<double*>(smem_raw);  // [nc*dim]
  int32_t* ids = reinterpret_cast<int32_t*>(Xs + nc * dim);  // [nn] (padded to even)
  double* pt_lds = reinterpret_cast<double*>(ids + ((nn + 1) & ~1));
  const int pt_stride = dd + 1 + dim * nn;  // per point: Jinv[dd], c, G[dim][nn]
  double* pt = PT_LDS ? pt_lds : (A.gscratch + (int64_t)blockIdx.x * A.gscratch_stride);

  const int tid = threadIdx.x;
  const int dw = dim == 2 ? 1 : 3;
  const int64_t e_end = DENSE ? 1 : A.n_elem;

  for (int64_t eq = blockIdx.x; eq < e_end; eq += gridDim.x) {
    const int64_t e = (!DENSE && A.esel) ? A.esel[eq] : eq;
    __syncthreads();
    // ---- gather corners + connectivity
    for (int t = tid; t < nn; t += BLOCK) ids[t] = DENSE ? t : A.conn[e * nn + t];
    for (int t = tid; t < nc * dim; t += BLOCK) {
      if (DENSE) {
        Xs[t] = A.corners[t];
      } else {
        int cn = t / dim, x = t - cn * dim;
        Xs[t] = A.xyz[(int64_t)A.conn[e * nn + cn] * dim + x];
      }
    }
    __syncthreads();
    // ---- geometry at every point: J = HrsCoo.X (spectral.py:120,140), inverse, c = w detJ (:122,142)
    for (int g = tid; g < ngt; g += BLOCK) {
      const int q = g < nga ? qa : 1;
      const int gl = g < nga ? g : g - nga;
      const double* hc = A.HrsCoo[q] + (int64_t)gl * dim * nc;
      double J[9], Ji[9];
      for (int d = 0; d < dim; ++d)
        for (int x = 0; x < dim; ++x) {
          double s = 0.0;
          for (int cn = 0; cn < nc; ++cn) s += hc[d * nc + cn] * Xs[cn * dim + x];
          J[d * dim + x] = s;
        }
      double det = inv_det(J, Ji, dim);
      double* P = pt + (int64_t)g * pt_stride;
      for (int i = 0; i < dd; ++i) P[i] = Ji[i];
      P[dd] = A.w[q][gl] * det;
    }
    __syncthreads();
    // ---- physical gradients G = J^-1 Hrs (spectral.py:121,141)
    for (int t = tid; t < ngt * dim * nn; t += BLOCK) {
      int g = t / (dim * nn);
      int r = t - g * dim * nn;
      int d = r / nn, a = r - d * nn;
      const int q = g < nga ? qa : 1;
      const int gl = g < nga ? g : g - nga;
      const double* hrs = A.Hrs[q] + (int64_t)gl * dim * nn;
      double* P = pt + (int64_t)g * pt_stride;
      double s = 0.0;
      for (int x = 0; x < dim; ++x) s += P[d * dim + x] * hrs[x * nn + a];
      P[dd + 1 + d * nn + a] = s;
    }
    __syncthreads();

    // ---- entries.  ab (node pair) is the fast index so that a thread keeps the same pair
    //      across components when nn*nn is a multiple of BLOCK (Q1 hex: 64 pairs = 64 lanes).
    const int npair = nn * nn;
    int last_ab = -1, slot = 0, r_lo = 0, r_len = 0, r_cr = 0, r_crfs = 0;   // r_cr / r_crfs: row start inside Krhs / Krhsfs
    auto locate = [&](int ab, int a, int b) {
      if (ab != last_ab) {
        last_ab = ab;
        int row = ids[a];
        if (row < A.n_owned) {
          r_lo = A.rowptr[row];
          r_len = A.rowptr[row + 1] - r_lo;
          slot = find_slot(A.colidx, r_lo, r_len, ids[b]);
          r_cr = A.rcrow ? A.rcrow[row] : r_lo;
          r_crfs = A.rcrow_fs ? A.rcrow_fs[row] : r_lo;
        } else {
          r_len = -1;
        }
      }
    };

    if (!PT_LDS && !DENSE && A.ho_chunk > 0) {
      // ---- high-order KLE elements: the point data (pt) sits in global scratch and an entry-per-lane loop re-reads
      //      it nn^2 times.  Here the points go through LDS `CH` at a time and every lane owns a 2x2 micro-block of
      //      node pairs whose 3x3 blocks accumulate in registers: 17 LDS reads per point serve 4 pairs.  Entries are
      //      summed in the order g = 0, 1, ... as everywhere else (bit-identical to the per-entry formulation).
      const int CH = A.ho_chunk;
      const int CS = (dim * nn + nn + 1 + 1) & ~1;                 // per point: G[dim][nn], H row [nn], c
      double* chunk = reinterpret_cast<double*>(ids + ((nn + 1) & ~1));
      const int nb2 = (nn + 1) / 2, n_mb = nb2 * nb2;
      int sgv[3][3], mv[3][3], sg2v[3][3], m2v[3][3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          sgv[p][k] = sg2v[p][k] = 0;
          mv[p][k] = m2v[p][k] = 0;
          if (p < dim && k < dw) {
            int m = 0;
            sgv[p][k] = curl_term(dim, p, k, &m);
            mv[p][k] = m;
            int m2 = 0;
            if (dim == 2) {
              m2 = 1 - p;
              sg2v[p][k] = p == 0 ? -1 : 1;
            } else {
              sg2v[p][k] = curl_term(3, k, p, &m2);
            }
            m2v[p][k] = m2;
          }
        }
      const bool want_k = A.K || A.Krhs, want_r = A.Rw || A.Rd;
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0 ? !want_k : !want_r) continue;
        for (int mb0 = 0; mb0 < n_mb; mb0 += BLOCK) {
          const int mb = mb0 + tid;
          const bool active = mb < n_mb;
          const int bi = active ? mb / nb2 : 0, bj = active ? mb - bi * nb2 : 0;
          const int an[2] = {2 * bi, min(2 * bi + 1, nn - 1)}, bn[2] = {2 * bj, min(2 * bj + 1, nn - 1)};
          const bool av[2] = {active, active && 2 * bi + 1 < nn}, bv[2] = {active, active && 2 * bj + 1 < nn};
          double acc[4][3][3], accd[4][3];
#pragma unroll
          for (int pr = 0; pr < 4; ++pr)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              accd[pr][p] = 0.0;
#pragma unroll
              for (int q = 0; q < 3; ++q) acc[pr][p][q] = 0.0;
            }
          for (int g0 = 0; g0 < ngt; g0 += CH) {
            const int ng = min(CH, ngt - g0);
            __syncthreads();
            for (int t = tid; t < ng * CS; t += BLOCK) {
              const int gg = t / CS, idx = t - gg * CS, g = g0 + gg;
              const double* P = pt + (int64_t)g * pt_stride;
              double val = 0.0;
              if (idx < dim * nn)
                val = P[dd + 1 + idx];
              else if (idx < dim * nn + nn)
                val = g < nga ? A.H[0][(int64_t)g * nn + (idx - dim * nn)] : A.H[1][(int64_t)(g - nga) * nn + (idx - dim * nn)];
              else if (idx == dim * nn + nn)
                val = P[dd];
              chunk[t] = val;
            }
            __syncthreads();
            if (!active) continue;
            for (int gg = 0; gg < ng; ++gg) {
              const double* Cg = chunk + gg * CS;
              const bool full = g0 + gg < nga;
              const double cg = Cg[dim * nn + nn];
              double ga[2][3], gb[2][3], ha[2], hb[2];
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                ha[u] = Cg[dim * nn + an[u]];
                hb[u] = Cg[dim * nn + bn[u]];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                  ga[u][d] = d < dim ? Cg[d * nn + an[u]] : 0.0;
                  gb[u][d] = d < dim ? Cg[d * nn + bn[u]] : 0.0;
                }
              }
#pragma unroll
              for (int ua = 0; ua < 2; ++ua)
#pragma unroll
                for (int ub = 0; ub < 2; ++ub) {
                  const int pr = ua * 2 + ub;
                  if (pass == 0) {
                    double sdot = 0.0;
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                      if (d < dim) sdot += ga[ua][d] * gb[ub][d];
                    if (full) {
#pragma unroll
                      for (int p = 0; p < 3; ++p)
                        if (p < dim) acc[pr][p][p] += cg * sdot;
                    } else {
#pragma unroll
                      for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                          if (p < dim && q < dim) {
                            double pen = A.alpha_d * ga[ua][p] * gb[ub][q] - A.alpha_w * ga[ua][q] * gb[ub][p];
                            if (p == q) pen += A.alpha_w * sdot;
                            acc[pr][p][q] += cg * pen;
                          }
                    }
                  } else if (full) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                      if (p < dim) accd[pr][p] -= cg * ha[ua] * gb[ub][p];
#pragma unroll
                      for (int k = 0; k < 3; ++k)
                        if (sgv[p][k] != 0) acc[pr][p][k] += cg * ha[ua] * (double)sgv[p][k] * pick3(gb[ub], mv[p][k]);
                    }
                  } else {
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                      if (p < dim) accd[pr][p] += cg * A.alpha_d * ga[ua][p] * hb[ub];
#pragma unroll
                      for (int k = 0; k < 3; ++k)
                        if (sgv[p][k] != 0) acc[pr][p][k] += cg * A.alpha_w * (double)sg2v[p][k] * pick3(ga[ua], m2v[p][k]) * hb[ub];
                    }
                  }
                }
            }
          }
          // ---- scatter the micro-block (same routing as the per-pair path)
#pragma unroll
          for (int ua = 0; ua < 2; ++ua)
#pragma unroll
            for (int ub = 0; ub < 2; ++ub) {
              if (!av[ua] || !bv[ub]) continue;
              const int pr = ua * 2 + ub, a = an[ua], b = bn[ub];
              const int row = ids[a];
              if (row >= A.n_owned) continue;
              const int lo = A.rowptr[row], len = A.rowptr[row + 1] - lo;
              const int sl = find_slot(A.colidx, lo, len, ids[b]);
#pragma unroll
              for (int p = 0; p < 3; ++p) {
                if (p >= dim) continue;
                const int64_t rd = (int64_t)row * dim + p;
                const int ci = A.bcmask ? A.bcmask[rd] : 0;
                if (pass == 0) {
#pragma unroll
                  for (int q = 0; q < 3; ++q) {
                    if (q >= dim) continue;
                    const int cj = A.bcmask ? A.bcmask[(int64_t)ids[b] * dim + q] : 0;
                    const int64_t off = ((int64_t)lo * dim + (int64_t)p * len + sl) * dim + q;
                    const double vv = acc[pr][p][q];
                    if (ci == 0) {
                      if (cj == 0) {
                        if (A.K) atomicAdd(&A.K[off], vv);
                      } else if (A.Krhs) {
                        const int cr = A.rcrow ? A.rcrow[row] : lo;
                        if (cr >= 0) atomicAdd(&A.Krhs[((int64_t)cr * dim + (int64_t)p * len + sl) * dim + q], -vv);
                      }
                    }
                  }
                } else {
                  if (A.Rw) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                      if (k >= dw) continue;
                      const double vv = acc[pr][p][k];
                      if (vv != 0.0 && ci == 0) atomicAdd(&A.Rw[((int64_t)lo * dim + (int64_t)p * len + sl) * dw + k], vv);
                    }
                  }
                  if (A.Rd && ci == 0) atomicAdd(&A.Rd[(int64_t)lo * dim + (int64_t)p * len + sl], accd[pr][p]);
                }
              }
            }
        }
      }
      continue;
    }

    if (oper) {
      const int br = A.op_br, bc = A.op_bc;
      for (int t = tid; t < npair * br * bc; t += BLOCK) {
        int pq = t / npair, ab = t - pq * npair;
        int a = ab / nn, b = ab - a * nn;
        int p = pq / bc, q = pq - p * bc;
        double v = 0.0;
        for (int g = 0; g < nga; ++g) {
          const double* P = pt + (int64_t)g * pt_stride;
          const double ha = A.H[qa][(int64_t)g * nn + a];
          if (ha == 0.0) continue;  // nodal rules: H is the identity
          double sterm = 0.0;
          for (int k = 0; k < A.op_nterms; ++k)
            if (A.op_terms[3 * k] == p && A.op_terms[3 * k + 1] == q)
              sterm += A.op_coef[k] * P[dd + 1 + A.op_terms[3 * k + 2] * nn + b];
          v += P[dd] * ha * sterm;
        }
        if (DENSE) {
          A.out0[(int64_t)(a * br + p) * (bc * nn) + b * bc + q] = v;
        } else {
          locate(ab, a, b);
          if (r_len < 0) continue;
          int64_t off = ((int64_t)r_lo * br + (int64_t)p * r_len + slot) * bc + q;
          if (v != 0.0) atomicAdd(&A.K[off], v);
        }
      }
      continue;
    }
    if (!kle) {
      // scalar forms: A_ab = sum c G_a.G_b (Laplace) or sum c H_a H_b (mass)
      for (int ab = tid; ab < npair; ab += BLOCK) {
        int a = ab / nn, b = ab - a * nn;
        double v = 0.0;
        for (int g = 0; g < nga; ++g) {
          const double* P = pt + (int64_t)g * pt_stride;
          double cg = P[dd];
          if (A.form == PYN_FORM_LAPLACE) {
            double s = 0.0;
            for (int d = 0; d < dim; ++d) s += P[dd + 1 + d * nn + a] * P[dd + 1 + d * nn + b];
            v += cg * s;
          } else {
            const double* Hq = A.H[qa] + (int64_t)g * nn;
            v += cg * Hq[a] * Hq[b];
          }
        }
        if (DENSE) {
          A.out0[ab] = v;
        } else {
          locate(ab, a, b);
          if (r_len < 0) continue;
          int ra = ids[a], cb = ids[b];
          bool mr = A.bcmask && A.bcmask[ra], mc = A.bcmask && A.bcmask[cb];
          if (mr) continue;
          int64_t off = (int64_t)r_lo + slot;
          if (mc) {
            if (A.Krhs && r_cr >= 0) atomicAdd(&A.Krhs[(int64_t)r_cr + slot], -v);
          } else if (A.K) {
            atomicAdd(&A.K[off], v);
          }
        }
      }
      continue;
    }

    // ---- KLE blocks.  One lane per NODE PAIR (a, b) computing all components of its block: the point data
    //      (c, G[.][a], G[.][b], H) is read once per pair instead of once per entry, which is what bounds the
    //      high-order elements whose point data lives in global scratch.  Every entry is still summed in the
    //      order g = 0, 1, ... of the per-entry formulation (bit-identical results).
    // ---- K (dim x dim blocks); component loops have the static bound 3 with `< dim` guards so that the
    //      accumulators stay in registers
    if (DENSE ? (A.out0 != nullptr) : (A.K != nullptr || A.Krhs != nullptr || A.Kfs != nullptr || A.Krhsfs != nullptr)) {
      for (int ab = tid; ab < npair; ab += BLOCK) {
        const int a = ab / nn, b = ab - a * nn;
        double vfull = 0.0;
        for (int g = 0; g < nga; ++g) {
          const double* P = pt + (int64_t)g * pt_stride;
          double s = 0.0;
          for (int d = 0; d < dim; ++d) s += P[dd + 1 + d * nn + a] * P[dd + 1 + d * nn + b];
          vfull += P[dd] * s;
        }
        double v[3][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int q = 0; q < 3; ++q) v[p][q] = p == q ? vfull : 0.0;
        for (int g = nga; g < ngt; ++g) {
          const double* P = pt + (int64_t)g * pt_stride;
          const double* G = P + dd + 1;
          double ga[3] = {0.0, 0.0, 0.0}, gb[3] = {0.0, 0.0, 0.0}, s = 0.0;
#pragma unroll
          for (int d = 0; d < 3; ++d)
            if (d < dim) {
              ga[d] = G[d * nn + a];
              gb[d] = G[d * nn + b];
              s += ga[d] * gb[d];
            }
          const double cg = P[dd];
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q)
              if (p < dim && q < dim) {
                double pen = A.alpha_d * ga[p] * gb[q] - A.alpha_w * ga[q] * gb[p];
                if (p == q) pen += A.alpha_w * s;
                v[p][q] += cg * pen;
              }
        }
        if (!DENSE) {
          locate(ab, a, b);
          if (r_len < 0) continue;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            if (p < dim && q < dim) {
              const double vv = v[p][q];
              if (DENSE) {
                A.out0[(int64_t)(a * dim + p) * (dim * nn) + b * dim + q] = vv;
              } else {
                int64_t rd = (int64_t)ids[a] * dim + p, cd = (int64_t)ids[b] * dim + q;
                const int ci = A.bcmask ? A.bcmask[rd] : 0, cj = A.bcmask ? A.bcmask[cd] : 0;
                int64_t off = ((int64_t)r_lo * dim + (int64_t)p * r_len + slot) * dim + q;
                if (ci == 0) {                                   // base_problem.py:426-427 / 388-390
                  if (cj == 0) {
                    if (A.K) atomicAdd(&A.K[off], vv);
                  } else if (A.Krhs && r_cr >= 0) {
                    atomicAdd(&A.Krhs[((int64_t)r_cr * dim + (int64_t)p * r_len + slot) * dim + q], -vv);
                  }
                }
                if (A.Kfs && ((ci == 1 && cj <= 1) || (ci == 0 && cj == 1))) atomicAdd(&A.Kfs[off], vv);   // :396-407
                if (A.Krhsfs && ci <= 1 && cj == 2 && r_crfs >= 0)                                         // :417-422
                  atomicAdd(&A.Krhsfs[((int64_t)r_crfs * dim + (int64_t)p * r_len + slot) * dim + q], -vv);
              }
            }
      }
    }
    // ---- Rw (dim x dim_w blocks)
    if (DENSE ? (A.out1 != nullptr) : (A.Rw != nullptr || A.Rwfs != nullptr)) {
      for (int ab = tid; ab < npair; ab += BLOCK) {
        const int a = ab / nn, b = ab - a * nn;
        double v[3][3];
        int sgv[3][3], mv[3][3], sg2v[3][3], m2v[3][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            v[p][k] = 0.0;
            sgv[p][k] = 0;
            mv[p][k] = m2v[p][k] = 0;
            sg2v[p][k] = 0;
            if (p < dim && k < dw) {
              int m = 0;
              sgv[p][k] = curl_term(dim, p, k, &m);  // (curl w)_p picks  sg * d_m w_k
              mv[p][k] = m;
              // reduced: alpha_w c Bc[k][(a,p)] H_b with (curl v)_k = eps_{k m p} d_m v_p
              int m2 = 0;
              if (dim == 2) {  // w = d_x v_y - d_y v_x
                m2 = 1 - p;
                sg2v[p][k] = p == 0 ? -1 : 1;
              } else {
                sg2v[p][k] = curl_term(3, k, p, &m2);
              }
              m2v[p][k] = m2;
            }
          }
        for (int g = 0; g < nga; ++g) {
          const double* P = pt + (int64_t)g * pt_stride;
          const double cg = P[dd], ha = A.H[0][(int64_t)g * nn + a];
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int k = 0; k < 3; ++k)
              if (sgv[p][k] != 0) v[p][k] += cg * ha * (double)sgv[p][k] * P[dd + 1 + mv[p][k] * nn + b];
        }
        for (int g = nga; g < ngt; ++g) {
          const double* P = pt + (int64_t)g * pt_stride;
          const double cg = P[dd], hb = A.H[1][(int64_t)(g - nga) * nn + b];
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int k = 0; k < 3; ++k)
              if (sgv[p][k] != 0) v[p][k] += cg * A.alpha_w * (double)sg2v[p][k] * P[dd + 1 + m2v[p][k] * nn + a] * hb;
        }
        if (!DENSE) {
          locate(ab, a, b);
          if (r_len < 0) continue;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (p < dim && k < dw) {
              const double vv = v[p][k];
              if (DENSE) {
                A.out1[(int64_t)(a * dim + p) * (dw * nn) + b * dw + k] = vv;
              } else {
                int64_t rd = (int64_t)ids[a] * dim + p;
                const int ci = A.bcmask ? A.bcmask[rd] : 0;
                int64_t off = ((int64_t)r_lo * dim + (int64_t)p * r_len + slot) * dw + k;
                if (vv != 0.0) {
                  if (ci == 0 && A.Rw) atomicAdd(&A.Rw[off], vv);
                  if (ci == 1 && A.Rwfs) atomicAdd(&A.Rwfs[off], vv);      // base_problem.py:412-413
                }
              }
            }
      }
    }
    // ---- Rd (dim x 1 blocks)
    if (DENSE ? (A.out2 != nullptr) : (A.Rd != nullptr || A.Rdfs != nullptr)) {
      for (int ab = tid; ab < npair; ab += BLOCK) {
        const int a = ab / nn, b = ab - a * nn;
        double v[3] = {0.0, 0.0, 0.0};
        for (int g = 0; g < nga; ++g) {
          const double* P = pt + (int64_t)g * pt_stride;
          const double cg = P[dd], ha = A.H[0][(int64_t)g * nn + a];
#pragma unroll
          for (int p = 0; p < 3; ++p)
            if (p < dim) v[p] -= cg * ha * P[dd + 1 + p * nn + b];
        }
        for (int g = nga; g < ngt; ++g) {
          const double* P = pt + (int64_t)g * pt_stride;
          const double cg = P[dd], hb = A.H[1][(int64_t)(g - nga) * nn + b];
#pragma unroll
          for (int p = 0; p < 3; ++p)
            if (p < dim) v[p] += cg * A.alpha_d * P[dd + 1 + p * nn + a] * hb;
        }
        if (!DENSE) {
          locate(ab, a, b);
          if (r_len < 0) continue;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
          if (p < dim) {
            if (DENSE) {
              A.out2[(int64_t)(a * dim + p) * nn + b] = v[p];
            } else {
              int64_t rd = (int64_t)ids[a] * dim + p;
              const int ci = A.bcmask ? A.bcmask[rd] : 0;
              int64_t off = (int64_t)r_lo * dim + (int64_t)p * r_len + slot;
              if (ci == 0 && A.Rd) atomicAdd(&A.Rd[off], v[p]);
              if (ci == 1 && A.Rdfs) atomicAdd(&A.Rdfs[off], v[p]);        // base_problem.py:415-416
            }
          }
      }
    }
  }
}

// Unit diagonal on imposed DOFs: Mat.setIndices2One (mat_generator.py:113-118), single-rank
// semantics (diag = 1).  One thread per owned DOF.  With the no-slip split: K, Krhs get 1 on every
// non-free DOF, Kfs gets -1 on the tangential no-slip DOFs (base_problem.py:441-442) so that K + Kfs
// keeps only the doubly imposed DOFs eliminated, Krhsfs gets 1 on those (:449-450).
__global__ void bc_identity_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                   const uint8_t* __restrict__ mask, int64_t n_owned, int ndof, double* __restrict__ K,
                                   double* __restrict__ Krhs, double* __restrict__ Kfs, double* __restrict__ Krhsfs,
                                   const int32_t* __restrict__ rcrow, const int32_t* __restrict__ rcrow_fs) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_owned * ndof; t += (int64_t)gridDim.x * blockDim.x) {
    const int cls = mask[t];
    if (!cls) continue;
    int64_t i = t / ndof;
    int p = (int)(t - i * ndof);
    int lo = rowptr[i], len = rowptr[i + 1] - lo;
    int slot = find_slot(colidx, lo, len, (int)i);
    int64_t off = ((int64_t)lo * ndof + (int64_t)p * len + slot) * ndof + p;
    if (K) K[off] = 1.0;
    if (Krhs) {
      const int cr = rcrow ? rcrow[i] : lo;      // (an imposed row is always stored by a compact matrix)
      if (cr >= 0) Krhs[((int64_t)cr * ndof + (int64_t)p * len + slot) * ndof + p] = 1.0;
    }
    if (Kfs && cls == 1) Kfs[off] -= 1.0;
    if (Krhsfs && cls == 2) {
      const int cr = rcrow_fs ? rcrow_fs[i] : lo;
      if (cr >= 0) Krhsfs[((int64_t)cr * ndof + (int64_t)p * len + slot) * ndof + p] = 1.0;
    }
  }
}

// Linear simplices (constant gradients): one element per lane, no LDS, no barriers.  L_ab = (sum_g w_g) detJ
// G_a.G_b with G = J^-1 Hrs; rows owned by this rank and not eliminated are scatter-added with FP64 atomics,
// the slot found by bisection in the (short: ~7 / ~15 entries) sorted row.  Same elimination rule as the
// generic kernel.  The irregular-indexing workload of BASELINE.json configs[4].
template <int DIM>
__global__ void __launch_bounds__(256) assemble_p1_laplace_kernel(const int32_t* __restrict__ conn, const double* __restrict__ xyz,
                                                                    int64_t n_elem, int64_t n_owned,
                                                                    const int32_t* __restrict__ rowptr,
                                                                    const int32_t* __restrict__ colidx,
                                                                    const uint8_t* __restrict__ bcmask,
                                                                    const double* __restrict__ hrs, double wsum,
                                                                    double* __restrict__ A, double* __restrict__ Arhs,
                                                                    const int32_t* __restrict__ rcrow) {
  constexpr int NN = DIM + 1;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_elem) return;
  int ids[NN];
  double X[NN][DIM];
  if (DIM == 3) {
    const int4 q = *reinterpret_cast<const int4*>(conn + e * 4);
    ids[0] = q.x, ids[1] = q.y, ids[2] = q.z, ids[NN - 1] = q.w;
  } else {
#pragma unroll
    for (int a = 0; a < NN; ++a) ids[a] = conn[e * NN + a];
  }
#pragma unroll
  for (int a = 0; a < NN; ++a)
#pragma unroll
    for (int d = 0; d < DIM; ++d) X[a][d] = xyz[(int64_t)ids[a] * DIM + d];
  double H[DIM][NN];
#pragma unroll
  for (int d = 0; d < DIM; ++d)
#pragma unroll
    for (int a = 0; a < NN; ++a) H[d][a] = hrs[d * NN + a];  // uniform: scalar loads
  double J[DIM * DIM], Ji[DIM * DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d)
#pragma unroll
    for (int x = 0; x < DIM; ++x) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < NN; ++a) s += H[d][a] * X[a][x];
      J[d * DIM + x] = s;
    }
  const double cw = wsum * inv_det(J, Ji, DIM);
  double G[DIM][NN];
#pragma unroll
  for (int x = 0; x < DIM; ++x)
#pragma unroll
    for (int a = 0; a < NN; ++a) {
      double s = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) s += Ji[x * DIM + d] * H[d][a];
      G[x][a] = s;
    }
  bool mc[NN];
#pragma unroll
  for (int a = 0; a < NN; ++a) mc[a] = bcmask && bcmask[ids[a]];
#pragma unroll
  for (int a = 0; a < NN; ++a) {
    if (ids[a] >= n_owned || mc[a]) continue;
    const int lo = rowptr[ids[a]], len = rowptr[ids[a] + 1] - lo;
#pragma unroll
    for (int b = 0; b < NN; ++b) {
      double v = 0.0;
#pragma unroll
      for (int x = 0; x < DIM; ++x) v += G[x][a] * G[x][b];
      v *= cw;
      const int sl = find_slot(colidx, lo, len, ids[b]);
      const int64_t off = (int64_t)lo + sl;
      if (mc[b]) {
        const int cr = rcrow ? rcrow[ids[a]] : lo;
        if (Arhs && cr >= 0) atomicAdd(&Arhs[(int64_t)cr + sl], -v);
      } else if (A) {
        atomicAdd(&A[off], v);
      }
    }
  }
}

// ---- High-order KLE elements on the FP64 matrix cores -----------------------------------------------------------
// For nn >~ 64 nodes per element the blocks are dense contractions over the Gauss points,
//   S_f  = sum_full c G_d[a] G_d[b]           T_pq = sum_red c G_p[a] G_q[b]
//   U_m  = sum_full c H[a]  G_m[b]            V_m  = sum_red c G_m[a] H[b]
//   K[p][q]  = d_pq (S_f + aw sum_d T_dd) + ad T_pq - aw T_qp                                (spectral.py:131,152-153)
//   Rw[p][k] = sg_pk U_m(p,k) + aw sg2_pk V_m2(p,k)      Rd[p] = -U_p + ad V_p               (:132-133,155-156)
// i.e. small GEMMs with the points as the K dimension: v_mfma_f64_16x16x4_f64 (lane l: A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15], D col = l&15, row = (l>>4) + 4 reg).  One workgroup per element, each of its 4 waves owns one
// 16x16 tile of node pairs per round with all 16 accumulator tiles in registers; the points pass through LDS in chunks.
typedef double pyn_d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) assemble_kle_ho_mfma_kernel(AsmArgs A, int CH) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int dim = A.dim, nn = A.nn, nc = A.nc, dd = dim * dim;
  const int nga = A.ngp[0], ngb = A.ngp[1], ngt = nga + ngb;
  const int nn16 = (nn + 15) & ~15, nt16 = nn16 >> 4;
  double* Xs = reinterpret_cast<double*>(smem_raw);                      // [nc*dim]
  int32_t* ids = reinterpret_cast<int32_t*>(Xs + nc * dim);             // [nn]
  double* Gs = reinterpret_cast<double*>(ids + ((nn + 1) & ~1));        // [dim][CH][nn16]
  double* Hs = Gs + (size_t)dim * CH * nn16;                             // [CH][nn16]
  double* cs = Hs + (size_t)CH * nn16;                                   // [CH]
  const int pt_stride = dd + 1 + dim * nn;
  double* pt = A.gscratch + (int64_t)blockIdx.x * A.gscratch_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dw = dim == 2 ? 1 : 3;
  const int li = lane & 15, lk = lane >> 4;

  int sgv[3][3], mv[3][3], sg2v[3][3], m2v[3][3];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      sgv[p][k] = sg2v[p][k] = 0;
      mv[p][k] = m2v[p][k] = 0;
      if (p < dim && k < dw) {
        int m = 0;
        sgv[p][k] = curl_term(dim, p, k, &m);
        mv[p][k] = m;
        int m2 = 0;
        if (dim == 2) {
          m2 = 1 - p;
          sg2v[p][k] = p == 0 ? -1 : 1;
        } else {
          sg2v[p][k] = curl_term(3, k, p, &m2);
        }
        m2v[p][k] = m2;
      }
    }

  for (int64_t eq = blockIdx.x; eq < A.n_elem; eq += gridDim.x) {
    const int64_t e = A.esel ? A.esel[eq] : eq;
    __syncthreads();
    for (int t = tid; t < nn; t += 256) ids[t] = A.conn[e * nn + t];
    for (int t = tid; t < nc * dim; t += 256) {
      int cn = t / dim, x = t - cn * dim;
      Xs[t] = A.xyz[(int64_t)A.conn[e * nn + cn] * dim + x];
    }
    __syncthreads();
    for (int g = tid; g < ngt; g += 256) {   // geometry at every point (spectral.py:120-122,140-142)
      const int q = g < nga ? 0 : 1;
      const int gl = g < nga ? g : g - nga;
      const double* hc = A.HrsCoo[q] + (int64_t)gl * dim * nc;
      double J[9], Ji[9];
      for (int d = 0; d < dim; ++d)
        for (int x = 0; x < dim; ++x) {
          double sacc = 0.0;
          for (int cn = 0; cn < nc; ++cn) sacc += hc[d * nc + cn] * Xs[cn * dim + x];
          J[d * dim + x] = sacc;
        }
      double det = inv_det(J, Ji, dim);
      double* P = pt + (int64_t)g * pt_stride;
      for (int i = 0; i < dd; ++i) P[i] = Ji[i];
      P[dd] = A.w[q][gl] * det;
    }
    __syncthreads();
    for (int t = tid; t < ngt * dim * nn; t += 256) {   // G = J^-1 Hrs (:121,141)
      int g = t / (dim * nn);
      int r = t - g * dim * nn;
      int d = r / nn, a = r - d * nn;
      const int q = g < nga ? 0 : 1;
      const int gl = g < nga ? g : g - nga;
      const double* hrs = A.Hrs[q] + (int64_t)gl * dim * nn;
      double* P = pt + (int64_t)g * pt_stride;
      double sacc = 0.0;
      for (int x = 0; x < dim; ++x) sacc += P[d * dim + x] * hrs[x * nn + a];
      P[dd + 1 + d * nn + a] = sacc;
    }
    __syncthreads();

    // ---- tiles of 16 x 16 node pairs, four at a time (one per wave)
    const int n_tiles = nt16 * nt16;
    for (int t0 = 0; t0 < n_tiles; t0 += 4) {
      const int tile = t0 + wave;
      const bool active = tile < n_tiles;
      const int ta = active ? tile / nt16 : 0, tb = active ? tile - ta * nt16 : 0;
      const int a0 = ta * 16, b0 = tb * 16;
      pyn_d4 S = {0, 0, 0, 0}, T[3][3], U[3], V[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        U[p] = S;
        V[p] = S;
#pragma unroll
        for (int q = 0; q < 3; ++q) T[p][q] = S;
      }
      for (int rule = 0; rule < 2; ++rule) {
        const int gbeg = rule == 0 ? 0 : nga, gend = rule == 0 ? nga : ngt;
        for (int g0 = gbeg; g0 < gend; g0 += CH) {
          __syncthreads();
          // stage CH points (zero-padded past the rule's end and past nn)
          for (int t = tid; t < CH * nn16; t += 256) {
            const int kk = t / nn16, a = t - kk * nn16, g = g0 + kk;
            const bool ok = g < gend && a < nn;
            const double* P = pt + (int64_t)(ok ? g : 0) * pt_stride;
#pragma unroll
            for (int d = 0; d < 3; ++d)
              if (d < dim) Gs[((size_t)d * CH + kk) * nn16 + a] = ok ? P[dd + 1 + d * nn + a] : 0.0;
            Hs[(size_t)kk * nn16 + a] = ok ? A.H[rule][(int64_t)(g - gbeg) * nn + a] : 0.0;
            if (a == 0) cs[kk] = g < gend ? P[dd] : 0.0;
          }
          __syncthreads();
          if (!active) continue;
          for (int k4 = 0; k4 < CH; k4 += 4) {
            const int kk = k4 + lk;
            const double cg = cs[kk];
            double ga[3], gb[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              ga[d] = d < dim ? Gs[((size_t)d * CH + kk) * nn16 + a0 + li] : 0.0;
              gb[d] = d < dim ? Gs[((size_t)d * CH + kk) * nn16 + b0 + li] : 0.0;
            }
            const double ha = Hs[(size_t)kk * nn16 + a0 + li], hb = Hs[(size_t)kk * nn16 + b0 + li];
            if (rule == 0) {
#pragma unroll
              for (int d = 0; d < 3; ++d)
                if (d < dim) {
                  S = __builtin_amdgcn_mfma_f64_16x16x4f64(cg * ga[d], gb[d], S, 0, 0, 0);
                  U[d] = __builtin_amdgcn_mfma_f64_16x16x4f64(cg * ha, gb[d], U[d], 0, 0, 0);
                }
            } else {
#pragma unroll
              for (int p = 0; p < 3; ++p)
                if (p < dim) {
                  V[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(cg * ga[p], hb, V[p], 0, 0, 0);
#pragma unroll
                  for (int q = 0; q < 3; ++q)
                    if (q < dim) T[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(cg * ga[p], gb[q], T[p][q], 0, 0, 0);
                }
            }
          }
        }
      }
      if (!active) continue;
      // ---- combine and scatter: lane holds rows (lane>>4) + 4 r, column lane&15 of the tile
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = a0 + lk + 4 * r, b = b0 + li;
        if (a >= nn || b >= nn) continue;
        const int row = ids[a];
        if (row >= A.n_owned) continue;
        const int lo = A.rowptr[row], len = A.rowptr[row + 1] - lo;
        const int sl = find_slot(A.colidx, lo, len, ids[b]);
        double tdd = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
          if (d < dim) tdd += T[d][d][r];
        const double diag = S[r] + A.alpha_w * tdd;
        double uu[3] = {U[0][r], U[1][r], U[2][r]}, vv3[3] = {V[0][r], V[1][r], V[2][r]};
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          if (p >= dim) continue;
          const int ci = A.bcmask ? A.bcmask[(int64_t)row * dim + p] : 0;
          if (ci != 0) continue;
          if (A.K || A.Krhs) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              if (q >= dim) continue;
              double v = A.alpha_d * T[p][q][r] - A.alpha_w * T[q][p][r];
              if (p == q) v += diag;
              const int cj = A.bcmask ? A.bcmask[(int64_t)ids[b] * dim + q] : 0;
              const int64_t off = ((int64_t)lo * dim + (int64_t)p * len + sl) * dim + q;
              if (cj == 0) {
                if (A.K) atomicAdd(&A.K[off], v);
              } else if (A.Krhs) {
                const int cr = A.rcrow ? A.rcrow[row] : lo;
                if (cr >= 0) atomicAdd(&A.Krhs[((int64_t)cr * dim + (int64_t)p * len + sl) * dim + q], -v);
              }
            }
          }
          if (A.Rw) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              if (k >= dw || sgv[p][k] == 0) continue;
              const double v = (double)sgv[p][k] * pick3(uu, mv[p][k]) + A.alpha_w * (double)sg2v[p][k] * pick3(vv3, m2v[p][k]);
              if (v != 0.0) atomicAdd(&A.Rw[((int64_t)lo * dim + (int64_t)p * len + sl) * dw + k], v);
            }
          }
          if (A.Rd) atomicAdd(&A.Rd[(int64_t)lo * dim + (int64_t)p * len + sl], A.alpha_d * vv3[p] - uu[p]);
        }
      }
    }
  }
}

size_t generic_smem(const pyn_ctx* c, int ngt, bool pt_lds) {
  size_t s = (size_t)c->nc * c->dim * sizeof(double) + (size_t)((c->nn + 1) & ~1) * sizeof(int32_t);
  if (pt_lds) s += (size_t)ngt * (c->dim * c->dim + 1 + c->dim * c->nn) * sizeof(double);
  return s;
}

int fill_args(pyn_ctx* c, AsmArgs& A, int form) {
  A.conn = c->d_conn;
  A.xyz = c->d_xyz;
  A.n_elem = c->n_elem;
  A.n_owned = c->n_owned;
  A.dim = c->dim;
  A.nn = c->nn;
  A.nc = c->nc;
  for (int q = 0; q < 3; ++q) {
    A.ngp[q] = c->quad[q].ngp;
    A.w[q] = c->quad[q].w;
    A.H[q] = c->quad[q].H;
    A.Hrs[q] = c->quad[q].Hrs;
    A.HrsCoo[q] = c->quad[q].HrsCoo;
  }
  A.rowptr = c->d_rowptr;
  A.colidx = c->d_colidx;
  A.bcmask = nullptr;
  A.form = form;
  A.alpha_d = A.alpha_w = 0.0;
  A.K = A.Krhs = A.Rw = A.Rd = nullptr;
  A.Kfs = A.Krhsfs = A.Rwfs = A.Rdfs = nullptr;
  A.rcrow = A.rcrow_fs = A.esel = nullptr;
  A.corners = nullptr;
  A.out0 = A.out1 = A.out2 = nullptr;
  A.gscratch = nullptr;
  A.gscratch_stride = 0;
  A.ho_chunk = 0;
  A.op_rule = A.op_br = A.op_bc = A.op_nterms = 0;
  A.op_terms = nullptr;
  A.op_coef = nullptr;
  if (form == PYN_FORM_OPERATOR) return PYN_OK;  // rule checked by the caller
  const int qa = form == PYN_FORM_MASS_NODAL ? 2 : 0;
  PYN_CHECK(c->quad[qa].ngp > 0, "element tables for rule %d not set", qa);
  if (form == PYN_FORM_KLE) PYN_CHECK(c->quad[1].ngp > 0, "reduced-rule tables not set");
  return PYN_OK;
}

template <bool DENSE>
int launch_generic(pyn_ctx* c, AsmArgs& A, int64_t n_work) {
  const int qa = A.form == PYN_FORM_OPERATOR ? A.op_rule : (A.form == PYN_FORM_MASS_NODAL ? 2 : 0);
  const int ngt = A.ngp[qa] + (A.form == PYN_FORM_KLE ? A.ngp[1] : 0);
  const size_t pt_bytes = (size_t)ngt * (c->dim * c->dim + 1 + c->dim * c->nn) * sizeof(double);
  const bool pt_lds = pt_bytes <= 40 * 1024;
  const bool small = c->nn <= 8;
  int grid = (int)std::min<int64_t>(n_work, small ? 256 * 32 : 256 * 4);
  if (!pt_lds) {
    A.gscratch_stride = (int64_t)(pt_bytes / sizeof(double));
    PYN_TRY(pyn_ensure_work(c, (size_t)grid * pt_bytes));
    A.gscratch = c->d_work;
  }
  size_t smem = generic_smem(c, ngt, pt_lds);
  // high-order KLE elements (point data in global scratch): stage the Gauss points through LDS, see the kernel
  if (!DENSE && !pt_lds && A.form == PYN_FORM_KLE && !A.Kfs && !A.Krhsfs && !A.Rwfs && !A.Rdfs && !getenv("PYNAMA_NO_HO")) {
    const size_t cs = (size_t)((c->dim * c->nn + c->nn + 1 + 1) & ~1) * sizeof(double);
    const int ch = (int)std::max<size_t>(1, std::min<size_t>(16, (48 * 1024) / cs));
    PYN_CHECK(smem + ch * cs <= 160 * 1024, "element too large for the high-order staging buffer");
    A.ho_chunk = ch;
    smem += ch * cs;
  }
  if (A.ho_chunk > 0 && c->nn >= 48 && !getenv("PYNAMA_NO_HO_MFMA")) {
    // dense enough for the FP64 matrix cores: largest chunk of points (multiple of 4) whose staging fits the LDS
    const size_t nn16 = (size_t)((c->nn + 15) & ~15);
    const size_t fixed = generic_smem(c, ngt, false);
    int ch = 0;
    for (int k = 16; k >= 4; k -= 4)
      if (fixed + ((size_t)(c->dim + 1) * k * nn16 + k) * sizeof(double) <= 96 * 1024) {
        ch = k;
        break;
      }
    if (ch > 0) {
      const size_t lds = fixed + ((size_t)(c->dim + 1) * ch * nn16 + ch) * sizeof(double);
      PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_kle_ho_mfma_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      assemble_kle_ho_mfma_kernel<<<grid, 256, lds, c->stream>>>(A, ch);
      PYN_HIP(hipGetLastError());
      return PYN_OK;
    }
  }
  if (smem > 64 * 1024) {
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(assemble_generic_kernel<256, false, DENSE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  }
  if (small) {
    assemble_generic_kernel<64, true, DENSE><<<grid, 64, smem, c->stream>>>(A);
  } else if (pt_lds) {
    assemble_generic_kernel<256, true, DENSE><<<grid, 256, smem, c->stream>>>(A);
  } else {
    assemble_generic_kernel<256, false, DENSE><<<grid, 256, smem, c->stream>>>(A);
  }
  PYN_HIP(hipGetLastError());
  return PYN_OK;
}

}  // namespace

int pyn_assemble_q1_tiled(pyn_ctx* c, int form, double alpha_d, double alpha_w, double* K, double* Krhs, double* Rw,
                          double* Rd, bool* handled);

// Does the imposed-column matrix `id` already hold zeros wherever the current Dirichlet set leaves zeros (DMat::rhs_clean)?  Read
// BEFORE mat_ptr marks the matrix as changing.
static bool rhs_is_clean(pyn_ctx* c, int id) {
  if (id < 0 || id >= (int)c->mats.size() || !c->mats[id].live || getenv("PYNAMA_RHS_FULL_WRITE")) return false;
  const int64_t s = c->mats[id].rhs_clean;
  return s == PYN_RHS_ANY || s == c->bc_stamp;
}

// crow != null: the argument may be a compact imposed-column matrix (Krhs, Krhsfs, Arhs); its row selection is brought to the current
// Dirichlet set (values zeroed when it had to be rebuilt) and handed back, null for a matrix with the graph's full pattern
static int mat_ptr(pyn_ctx* c, int id, int br, int bc, const char* name, double** out, const int32_t** crow = nullptr) {
  *out = nullptr;
  if (crow) *crow = nullptr;
  if (id < 0) return PYN_OK;
  PYN_TRY(pyn_check_mat(c, id, name));
  DMat& m = c->mats[id];
  PYN_CHECK(m.br == br && m.bc == bc, "%s must have block shape %dx%d (has %dx%d)", name, br, bc, m.br, m.bc);
  PYN_CHECK(crow || !m.rhs_compact, "%s: a compact imposed-column matrix (pyn_mat_create_rhs) can only be the Krhs / Arhs target", name);
  if (m.rhs_compact) {
    PYN_TRY(pyn_rhs_ensure(c, m, true));   // laid out for the Dirichlet set of THIS assembly
    *crow = m.c_crow;
  }
  m.touch();  // values are about to change
  *out = m.val;
  return PYN_OK;
}

// blocks stored by the matrix behind a raw value pointer of this assembly (the compact Krhs target, or the graph's count)
static int64_t rhs_blocks(const pyn_ctx* c, int id) { return id >= 0 ? pyn_mat_blocks(c, c->mats[id]) : c->nnzb; }

static int run_assembly(pyn_ctx* c, int form, double alpha_d, double alpha_w, double* K, double* Krhs, double* Rw,
                        double* Rd, int variant, int64_t krhs_blocks) {
  PYN_CHECK(c->d_rowptr, "pyn_csr_symbolic first");
  const int ndof = form == PYN_FORM_KLE ? c->dim : 1;
  if (c->d_bcmask) PYN_CHECK(c->bc_ndof == ndof, "bc mask has ndof=%d, form needs %d", c->bc_ndof, ndof);
  AsmArgs A;
  PYN_TRY(fill_args(c, A, form));
  A.bcmask = c->d_bcmask;
  A.alpha_d = alpha_d;
  A.alpha_w = alpha_w;
  A.K = K;
  A.Krhs = Krhs;
  A.Rw = Rw;
  A.Rd = Rd;
  A.rcrow = c->asm_rcrow;
  PYN_HIP(hipEventRecord(c->ev0, c->stream));
  bool handled = false;
  c->asm_krhs_pending = false;
  if (variant != 0) PYN_TRY(pyn_assemble_q1_tiled(c, form, alpha_d, alpha_w, K, Krhs, Rw, Rd, &handled));
  if (handled && c->asm_krhs_pending) {
    // the kernel family that took K cannot address a compact Krhs: -K_e[free, bc] comes from the elements that hold an imposed node
    // (a few per cent of the mesh) through the generic kernel, into the zeroed compact matrix
    PYN_TRY(pyn_bc_elements(c));
    PYN_HIP(hipMemsetAsync(Krhs, 0, (size_t)krhs_blocks * ndof * ndof * sizeof(double), c->stream));
    if (c->n_esel > 0) {
      AsmArgs B = A;
      B.K = B.Rw = B.Rd = nullptr;
      B.esel = c->d_esel;
      B.n_elem = c->n_esel;
      PYN_TRY(launch_generic<false>(c, B, c->n_esel));
    }
    if (c->d_bcmask) {
      int64_t n = c->n_owned * ndof;
      int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
      bc_identity_kernel<<<grid, 256, 0, c->stream>>>(c->d_rowptr, c->d_colidx, c->d_bcmask, c->n_owned, ndof, nullptr, Krhs, nullptr, nullptr,
                                                    c->asm_rcrow, nullptr);
    }
    c->asm_krhs_pending = false;
  }
  if (!handled) {  // scatter-add path: values start from zero (the tiled path writes every entry itself)
    const int dw = c->dim == 2 ? 1 : 3;
    const size_t nb = (size_t)c->nnzb * sizeof(double);
    if (K) PYN_HIP(hipMemsetAsync(K, 0, nb * ndof * ndof, c->stream));
    if (Krhs) PYN_HIP(hipMemsetAsync(Krhs, 0, (size_t)krhs_blocks * ndof * ndof * sizeof(double), c->stream));
    if (Rw) PYN_HIP(hipMemsetAsync(Rw, 0, nb * ndof * dw, c->stream));
    if (Rd) PYN_HIP(hipMemsetAsync(Rd, 0, nb * ndof, c->stream));
    const QuadTab& q0 = c->quad[0];
    if (form == PYN_FORM_LAPLACE && c->nn == c->dim + 1 && q0.const_grad && !getenv("PYNAMA_NO_P1")) {
      const int grid = (int)((c->n_elem + 255) / 256);
      if (c->dim == 3)
        assemble_p1_laplace_kernel<3><<<grid, 256, 0, c->stream>>>(c->d_conn, c->d_xyz, c->n_elem, c->n_owned, c->d_rowptr,
                                                                  c->d_colidx, c->d_bcmask, q0.Hrs, q0.wsum, K, Krhs, c->asm_rcrow);
      else
        assemble_p1_laplace_kernel<2><<<grid, 256, 0, c->stream>>>(c->d_conn, c->d_xyz, c->n_elem, c->n_owned, c->d_rowptr,
                                                                  c->d_colidx, c->d_bcmask, q0.Hrs, q0.wsum, K, Krhs, c->asm_rcrow);
      PYN_HIP(hipGetLastError());
    } else {
      PYN_TRY(launch_generic<false>(c, A, c->n_elem));
    }
  }
  if (!handled && c->d_bcmask && (K || Krhs)) {
    int64_t n = c->n_owned * ndof;
    int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
    bc_identity_kernel<<<grid, 256, 0, c->stream>>>(c->d_rowptr, c->d_colidx, c->d_bcmask, c->n_owned, ndof, K, Krhs, nullptr, nullptr,
                                                    c->asm_rcrow, nullptr);
  }
  PYN_HIP(hipEventRecord(c->ev1, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  float ms = 0;
  PYN_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->timers[PYN_T_ASSEMBLE] = ms;
  return PYN_OK;
}

extern "C" int pyn_assemble_kle(pyn_ctx* c, double alpha_d, double alpha_w, int K, int Krhs, int Rw, int Rd, int variant) {
  PYN_CHECK(c, "ctx is NULL");
  PYN_HIP(hipSetDevice(c->device));
  const int dim = c->dim, dw = dim == 2 ? 1 : 3;
  double *pK, *pKr, *pRw, *pRd;
  PYN_TRY(mat_ptr(c, K, dim, dim, "K", &pK));
  c->asm_rhs_clean = Krhs != K && rhs_is_clean(c, Krhs);   // (read before mat_ptr marks the matrix as changing)
  PYN_TRY(mat_ptr(c, Krhs, dim, dim, "Krhs", &pKr, &c->asm_rcrow));
  if (c->asm_rcrow) c->asm_rhs_clean = false;              // a compact matrix has no zero blocks to skip: every stored row is written
  PYN_TRY(mat_ptr(c, Rw, dim, dw, "Rw", &pRw));
  PYN_TRY(mat_ptr(c, Rd, dim, 1, "Rd", &pRd));
  const int rc = run_assembly(c, PYN_FORM_KLE, alpha_d, alpha_w, pK, pKr, pRw, pRd, variant, rhs_blocks(c, Krhs));
  c->asm_rhs_clean = false;
  c->asm_rcrow = nullptr;
  if (rc == PYN_OK && pKr && pK) c->mats[Krhs].rhs_clean = c->bc_stamp;   // exactly the imposed-column matrix of this Dirichlet set
  return rc;
}

extern "C" int pyn_assemble_kle_noslip(pyn_ctx* c, double alpha_d, double alpha_w, const int* mat_ids /*[8]*/) {
  PYN_CHECK(c && mat_ids, "NULL argument");
  PYN_CHECK(c->d_rowptr, "pyn_csr_symbolic first");
  PYN_CHECK(c->d_bcmask && c->bc_ndof == c->dim, "pyn_bc_set(dim, classes) first: 0 free, 1 tangential no-slip, 2 imposed");
  PYN_HIP(hipSetDevice(c->device));
  const int dim = c->dim, dw = dim == 2 ? 1 : 3;
  const int shapes[8][2] = {{dim, dim}, {dim, dim}, {dim, dw}, {dim, 1}, {dim, dim}, {dim, dim}, {dim, dw}, {dim, 1}};
  const char* names[8] = {"K", "Krhs", "Rw", "Rd", "Kfs", "Krhsfs", "Rwfs", "Rdfs"};
  double* ptr[8];
  const int32_t* crow[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int k = 0; k < 8; ++k) {
    const bool rhs = k == 1 || k == 5;       // Krhs, Krhsfs may be compact imposed-column matrices
    PYN_TRY(mat_ptr(c, mat_ids[k], shapes[k][0], shapes[k][1], names[k], &ptr[k], rhs ? &crow[k] : nullptr));
    if (ptr[k])
      PYN_HIP(hipMemsetAsync(ptr[k], 0, (size_t)rhs_blocks(c, mat_ids[k]) * shapes[k][0] * shapes[k][1] * sizeof(double), c->stream));
  }
  AsmArgs A;
  PYN_TRY(fill_args(c, A, PYN_FORM_KLE));
  A.bcmask = c->d_bcmask;
  A.alpha_d = alpha_d;
  A.alpha_w = alpha_w;
  A.K = ptr[0];
  A.Krhs = ptr[1];
  A.Rw = ptr[2];
  A.Rd = ptr[3];
  A.Kfs = ptr[4];
  A.Krhsfs = ptr[5];
  A.Rwfs = ptr[6];
  A.Rdfs = ptr[7];
  A.rcrow = crow[1];
  A.rcrow_fs = crow[5];
  PYN_HIP(hipEventRecord(c->ev0, c->stream));
  PYN_TRY(launch_generic<false>(c, A, c->n_elem));
  int64_t n = c->n_owned * dim;
  int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
  bc_identity_kernel<<<grid, 256, 0, c->stream>>>(c->d_rowptr, c->d_colidx, c->d_bcmask, c->n_owned, dim, A.K, A.Krhs, A.Kfs, A.Krhsfs, A.rcrow,
                                                A.rcrow_fs);
  PYN_HIP(hipEventRecord(c->ev1, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  float ms = 0;
  PYN_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->timers[PYN_T_ASSEMBLE] = ms;
  return PYN_OK;
}

extern "C" int pyn_assemble_scalar(pyn_ctx* c, int form, int Aid, int Arhs, int variant) {
  PYN_CHECK(c, "ctx is NULL");
  PYN_CHECK(form == PYN_FORM_LAPLACE || form == PYN_FORM_MASS_NODAL || form == PYN_FORM_MASS_FULL, "bad scalar form");
  PYN_HIP(hipSetDevice(c->device));
  double *pA, *pAr;
  c->asm_rhs_clean = Arhs != Aid && rhs_is_clean(c, Arhs);
  PYN_TRY(mat_ptr(c, Aid, 1, 1, "A", &pA));
  PYN_TRY(mat_ptr(c, Arhs, 1, 1, "Arhs", &pAr, &c->asm_rcrow));
  if (c->asm_rcrow) c->asm_rhs_clean = false;
  // the Jacobi data of A: kernels that see whole rows (lattice store phases) write 1 / diagonal on the way out
  c->asm_dinv = nullptr;
  c->asm_dinv_written = false;
  if (Aid >= 0 && form == PYN_FORM_LAPLACE && !getenv("PYNAMA_NO_ASM_DINV")) {
    DMat& m = c->mats[Aid];
    if (!m.dinv) PYN_HIP(hipMalloc((void**)&m.dinv, (size_t)c->n_owned * sizeof(double)));
    c->asm_dinv = m.dinv;
  }
  const int rc = run_assembly(c, form, 0.0, 0.0, pA, pAr, nullptr, nullptr, variant, rhs_blocks(c, Arhs));
  if (rc == PYN_OK && c->asm_dinv && c->asm_dinv_written) c->mats[Aid].dinv_valid = true;
  c->asm_dinv = nullptr;
  c->asm_rhs_clean = false;
  c->asm_rcrow = nullptr;
  if (rc == PYN_OK && pAr && pA) c->mats[Arhs].rhs_clean = c->bc_stamp;
  return rc;
}

extern "C" int pyn_elem_local(pyn_ctx* c, int form, double alpha_d, double alpha_w, const double* corners, double* out0,
                              double* out1, double* out2) {
  PYN_CHECK(c && corners, "NULL argument");
  PYN_CHECK(c->nn > 0, "pyn_mesh_set first (it fixes dim / nn)");
  PYN_HIP(hipSetDevice(c->device));
  const int dim = c->dim, nn = c->nn, dw = dim == 2 ? 1 : 3;
  const bool kle = form == PYN_FORM_KLE;
  size_t n0 = kle ? (size_t)dim * nn * dim * nn : (size_t)nn * nn;
  size_t n1 = kle ? (size_t)dim * nn * dw * nn : 0;
  size_t n2 = kle ? (size_t)dim * nn * nn : 0;
  size_t ncor = (size_t)c->nc * dim;
  size_t need = (n0 + n1 + n2 + ncor) * sizeof(double);
  if (need > c->eloc_bytes) {
    if (c->d_eloc) PYN_HIP(hipFree(c->d_eloc));
    c->d_eloc = nullptr;
    PYN_HIP(hipMalloc((void**)&c->d_eloc, need));
    c->eloc_bytes = need;
  }
  double* d_cor = c->d_eloc;
  double* d0 = d_cor + ncor;
  double* d1 = d0 + n0;
  double* d2 = d1 + n1;
  PYN_HIP(hipMemcpyAsync(d_cor, corners, ncor * sizeof(double), hipMemcpyHostToDevice, c->stream));
  AsmArgs A;
  PYN_TRY(fill_args(c, A, form));
  A.alpha_d = alpha_d;
  A.alpha_w = alpha_w;
  A.corners = d_cor;
  A.out0 = out0 ? d0 : nullptr;
  A.out1 = (kle && out1) ? d1 : nullptr;
  A.out2 = (kle && out2) ? d2 : nullptr;
  PYN_TRY(launch_generic<true>(c, A, 1));
  if (out0) PYN_HIP(hipMemcpyAsync(out0, d0, n0 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (kle && out1) PYN_HIP(hipMemcpyAsync(out1, d1, n1 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (kle && out2) PYN_HIP(hipMemcpyAsync(out2, d2, n2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  return PYN_OK;
}

// ---- first-order operator blocks (SrT / DivSrT / Curl of Spectral.getElemKLEOperators, spectral.py:159-218)
static int upload_terms(pyn_ctx* c, int nterms, const int32_t* terms, const double* coef, int32_t** dt, double** dc) {
  PYN_HIP(hipMalloc((void**)dt, (size_t)nterms * 3 * sizeof(int32_t)));
  PYN_HIP(hipMalloc((void**)dc, (size_t)nterms * sizeof(double)));
  PYN_HIP(hipMemcpyAsync(*dt, terms, (size_t)nterms * 3 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  PYN_HIP(hipMemcpyAsync(*dc, coef, (size_t)nterms * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return PYN_OK;
}

static int check_operator(pyn_ctx* c, int rule, int br, int bc, int nterms, const int32_t* terms, const double* coef) {
  PYN_CHECK(c && terms && coef, "NULL argument");
  PYN_CHECK(rule >= 0 && rule < 3 && c->quad[rule].ngp > 0, "element tables for rule %d not set", rule);
  PYN_CHECK(br >= 1 && br <= 6 && bc >= 1 && bc <= 6 && nterms >= 1 && nterms <= 64, "bad operator shape");
  for (int k = 0; k < nterms; ++k)
    PYN_CHECK(terms[3 * k] >= 0 && terms[3 * k] < br && terms[3 * k + 1] >= 0 && terms[3 * k + 1] < bc && terms[3 * k + 2] >= 0 &&
                  terms[3 * k + 2] < c->dim,
              "operator term %d out of range", k);
  return PYN_OK;
}

extern "C" int pyn_assemble_operator(pyn_ctx* c, int rule, int nterms, const int32_t* terms, const double* coef, int mat_id) {
  PYN_TRY(pyn_check_mat(c, mat_id, "pyn_assemble_operator"));
  DMat& M = c->mats[mat_id];
  PYN_CHECK(!M.rhs_compact, "pyn_assemble_operator: the target must have the graph's full pattern");
  PYN_TRY(check_operator(c, rule, M.br, M.bc, nterms, terms, coef));
  PYN_HIP(hipSetDevice(c->device));
  int32_t* dt = nullptr;
  double* dc = nullptr;
  PYN_TRY(upload_terms(c, nterms, terms, coef, &dt, &dc));
  AsmArgs A;
  PYN_TRY(fill_args(c, A, PYN_FORM_OPERATOR));
  A.op_rule = rule;
  A.op_br = M.br;
  A.op_bc = M.bc;
  A.op_nterms = nterms;
  A.op_terms = dt;
  A.op_coef = dc;
  A.K = M.val;
  M.touch();
  PYN_HIP(hipEventRecord(c->ev0, c->stream));
  bool handled = false;   // structured meshes of parallelepipeds: the row-run kernels (no atomics, every row written once)
  PYN_TRY(pyn_assemble_ho3_operator(c, rule, M.br, M.bc, nterms, terms, coef, M.val, &handled));
  if (!handled) {
    PYN_HIP(hipMemsetAsync(M.val, 0, (size_t)c->nnzb * M.br * M.bc * sizeof(double), c->stream));
    PYN_TRY(launch_generic<false>(c, A, c->n_elem));
  }
  PYN_HIP(hipEventRecord(c->ev1, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  float ms = 0;
  PYN_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->timers[PYN_T_ASSEMBLE] = ms;
  PYN_HIP(hipFree(dt));
  PYN_HIP(hipFree(dc));
  return PYN_OK;
}

extern "C" int pyn_elem_operator_local(pyn_ctx* c, int rule, int br, int bc, int nterms, const int32_t* terms,
                                       const double* coef, const double* corners, double* out) {
  PYN_CHECK(corners && out, "NULL argument");
  PYN_TRY(check_operator(c, rule, br, bc, nterms, terms, coef));
  PYN_HIP(hipSetDevice(c->device));
  const size_t n0 = (size_t)br * c->nn * bc * c->nn, ncor = (size_t)c->nc * c->dim;
  const size_t need = (n0 + ncor) * sizeof(double);
  if (need > c->eloc_bytes) {
    if (c->d_eloc) PYN_HIP(hipFree(c->d_eloc));
    c->d_eloc = nullptr;
    PYN_HIP(hipMalloc((void**)&c->d_eloc, need));
    c->eloc_bytes = need;
  }
  int32_t* dt = nullptr;
  double* dc = nullptr;
  PYN_TRY(upload_terms(c, nterms, terms, coef, &dt, &dc));
  PYN_HIP(hipMemcpyAsync(c->d_eloc, corners, ncor * sizeof(double), hipMemcpyHostToDevice, c->stream));
  AsmArgs A;
  PYN_TRY(fill_args(c, A, PYN_FORM_OPERATOR));
  A.op_rule = rule;
  A.op_br = br;
  A.op_bc = bc;
  A.op_nterms = nterms;
  A.op_terms = dt;
  A.op_coef = dc;
  A.corners = c->d_eloc;
  A.out0 = c->d_eloc + ncor;
  PYN_TRY(launch_generic<true>(c, A, 1));
  PYN_HIP(hipMemcpyAsync(out, A.out0, n0 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  PYN_HIP(hipFree(dt));
  PYN_HIP(hipFree(dc));
  return PYN_OK;
}
