/* CPU oracle, C restatement (OpenMP) of the Pynama hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Used (a) by tests/ to cross-check the numpy oracle at larger sizes and (b) by bench.py's
 * `cpu_baseline` leg ("kind": "port") as the CPU rate reported next to the GPU numbers.
 * Nothing under pynama_amd/ may link or call it.
 *
 * It restates, loop for loop, the reference algorithm (paths relative to /root/reference/):
 *   orc_elem_kle      src/elements/spectral.py:89-157   (B-matrix products, full + reduced rule)
 *   orc_elem_laplace  spectral.py:117-131 restricted to one velocity component
 *   orc_csr_pattern   src/domain/dmplex.py:305-333      (node adjacency, sorted rows)
 *   orc_assemble_*    src/cases/base_problem.py:499-552 + src/matrices/mat_generator.py:113-118
 *   orc_pcg           PETSc KSPCG + PCJACOBI semantics behind src/solver/ksp_solver.py:9-19
 * Parity: pinned through tests/test_oracle_c.py against oracle/fem_oracle.py, which is pinned
 * against the reference's golden vectors.
 *
 * Block-CSR layout identical to include/pynama_hip.h:
 *   val[(rowptr[i]*br + p*len_i + k)*bc + q]
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXD 3

void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static double inv_det(const double* J, double* Ji, int dim) {
  if (dim == 2) {
    double det = J[0] * J[3] - J[1] * J[2];
    Ji[0] = J[3] / det; Ji[1] = -J[1] / det; Ji[2] = -J[2] / det; Ji[3] = J[0] / det;
    return det;
  }
  double c0 = J[4] * J[8] - J[5] * J[7], c1 = J[5] * J[6] - J[3] * J[8], c2 = J[3] * J[7] - J[4] * J[6];
  double det = J[0] * c0 + J[1] * c1 + J[2] * c2;
  Ji[0] = c0 / det; Ji[1] = (J[2] * J[7] - J[1] * J[8]) / det; Ji[2] = (J[1] * J[5] - J[2] * J[4]) / det;
  Ji[3] = c1 / det; Ji[4] = (J[0] * J[8] - J[2] * J[6]) / det; Ji[5] = (J[2] * J[3] - J[0] * J[5]) / det;
  Ji[6] = c2 / det; Ji[7] = (J[1] * J[6] - J[0] * J[7]) / det; Ji[8] = (J[0] * J[4] - J[1] * J[3]) / det;
  return det;
}

/* G = inv(HrsCoo.X) Hrs ; returns detJ.  spectral.py:120-122 */
static double point_geom(int dim, int nn, int nc, const double* hrscoo, const double* hrs, const double* X, double* G) {
  double J[9], Ji[9];
  for (int d = 0; d < dim; ++d)
    for (int x = 0; x < dim; ++x) {
      double s = 0;
      for (int c = 0; c < nc; ++c) s += hrscoo[d * nc + c] * X[c * dim + x];
      J[d * dim + x] = s;
    }
  double det = inv_det(J, Ji, dim);
  for (int d = 0; d < dim; ++d)
    for (int a = 0; a < nn; ++a) {
      double s = 0;
      for (int x = 0; x < dim; ++x) s += Ji[d * dim + x] * hrs[x * nn + a];
      G[d * nn + a] = s;
    }
  return det;
}

/* curl tables of spectral.py:26-33: {row, comp, deriv}, sign = (-1)^i */
static const int CW2[2][3] = {{0, 0, 1}, {1, 0, 0}};
static const int CV2[2][3] = {{0, 1, 0}, {0, 0, 1}};
static const int C3[6][3] = {{0, 2, 1}, {0, 1, 2}, {1, 0, 2}, {1, 2, 0}, {2, 1, 0}, {2, 0, 1}};

/* K_e [nd x nd], Rw_e [nd x dw nn], Rd_e [nd x nn], nd = dim nn.  scratch: >= (dim*dim+dim+dw+1)*nd*... */
void orc_elem_kle(int dim, int nn, int ngf, const double* wf, const double* Hf, const double* Hrsf, const double* Hcoof,
                  int ngr, const double* wr, const double* Hr, const double* Hrsr, const double* Hcoor, const double* X,
                  double alpha_d, double alpha_w, double* K, double* Rw, double* Rd) {
  const int nc = 1 << dim, nd = dim * nn, dw = dim == 2 ? 1 : 3, nw = dw * nn;
  double* G = (double*)malloc(sizeof(double) * dim * nn);
  double* Bgr = (double*)calloc((size_t)dim * dim * nd, sizeof(double));
  double* Hvel = (double*)calloc((size_t)dim * nd, sizeof(double));
  double* Bw = (double*)calloc((size_t)dim * nw, sizeof(double));
  double* Bdiv = (double*)calloc((size_t)nd, sizeof(double));
  double* Bcurl = (double*)calloc((size_t)dw * nd, sizeof(double));
  double* Hw = (double*)calloc((size_t)dw * nw, sizeof(double));
  memset(K, 0, sizeof(double) * nd * nd);
  if (Rw) memset(Rw, 0, sizeof(double) * nd * nw);
  if (Rd) memset(Rd, 0, sizeof(double) * nd * nn);
  const int ncurl = dim == 2 ? 2 : 6;
  for (int g = 0; g < ngf; ++g) {
    const double* H = Hf + (size_t)g * nn;
    double det = point_geom(dim, nn, nc, Hcoof + (size_t)g * dim * nc, Hrsf + (size_t)g * dim * nn, X, G);
    double c = wf[g] * det;
    for (int n = 0; n < dim; ++n)                                    /* :124-126 */
      for (int d = 0; d < dim; ++d)
        for (int a = 0; a < nn; ++a) {
          Bgr[(size_t)(dim * n + d) * nd + a * dim + n] = G[d * nn + a];
          if (d == 0) Hvel[(size_t)n * nd + a * dim + n] = H[a];
        }
    for (int i = 0; i < ncurl; ++i) {                                 /* :128-129 */
      const int* t = dim == 2 ? CW2[i] : C3[i];
      double s = (i % 2) ? -1.0 : 1.0;
      for (int a = 0; a < nn; ++a) Bw[(size_t)t[0] * nw + a * dw + t[1]] = s * G[t[2] * nn + a];
    }
    for (int i = 0; i < nd; ++i)
      for (int j = 0; j < nd; ++j) {                                  /* :131 */
        double s = 0;
        for (int k = 0; k < dim * dim; ++k) s += Bgr[(size_t)k * nd + i] * Bgr[(size_t)k * nd + j];
        K[(size_t)i * nd + j] += c * s;
      }
    if (Rw)
      for (int i = 0; i < nd; ++i)
        for (int j = 0; j < nw; ++j) {                                /* :132 */
          double s = 0;
          for (int k = 0; k < dim; ++k) s += Hvel[(size_t)k * nd + i] * Bw[(size_t)k * nw + j];
          Rw[(size_t)i * nw + j] += c * s;
        }
    if (Rd)
      for (int i = 0; i < nd; ++i)
        for (int j = 0; j < nn; ++j) {                                /* :133 */
          double s = 0;
          for (int k = 0; k < dim; ++k) s += Hvel[(size_t)k * nd + i] * G[k * nn + j];
          Rd[(size_t)i * nn + j] -= c * s;
        }
  }
  for (int g = 0; g < ngr; ++g) {
    const double* H = Hr + (size_t)g * nn;
    double det = point_geom(dim, nn, nc, Hcoor + (size_t)g * dim * nc, Hrsr + (size_t)g * dim * nn, X, G);
    double c = wr[g] * det;
    for (int n = 0; n < dim; ++n)
      for (int a = 0; a < nn; ++a) Bdiv[a * dim + n] = G[n * nn + a];  /* :145 */
    for (int i = 0; i < ncurl; ++i) {                                 /* :147-148 */
      const int* t = dim == 2 ? CV2[i] : C3[i];
      double s = (i % 2) ? -1.0 : 1.0;
      for (int a = 0; a < nn; ++a) Bcurl[(size_t)t[0] * nd + a * dim + t[1]] = s * G[t[2] * nn + a];
    }
    for (int n = 0; n < dw; ++n)
      for (int a = 0; a < nn; ++a) Hw[(size_t)n * nw + a * dw + n] = H[a];   /* :150 */
    for (int i = 0; i < nd; ++i)
      for (int j = 0; j < nd; ++j) {                                  /* :152-153 */
        double s = alpha_d * Bdiv[i] * Bdiv[j];
        for (int k = 0; k < dw; ++k) s += alpha_w * Bcurl[(size_t)k * nd + i] * Bcurl[(size_t)k * nd + j];
        K[(size_t)i * nd + j] += c * s;
      }
    if (Rw)
      for (int i = 0; i < nd; ++i)
        for (int j = 0; j < nw; ++j) {                                /* :155 */
          double s = 0;
          for (int k = 0; k < dw; ++k) s += Bcurl[(size_t)k * nd + i] * Hw[(size_t)k * nw + j];
          Rw[(size_t)i * nw + j] += c * alpha_w * s;
        }
    if (Rd)
      for (int a = 0; a < nn; ++a)
        for (int p = 0; p < dim; ++p)
          for (int b = 0; b < nn; ++b)                                 /* :156, flatten('F') */
            Rd[(size_t)(a * dim + p) * nn + b] += c * alpha_d * G[p * nn + a] * H[b];
  }
  free(G); free(Bgr); free(Hvel); free(Bw); free(Bdiv); free(Bcurl); free(Hw);
}

void orc_elem_laplace(int dim, int nn, int ng, const double* w, const double* Hrs, const double* Hcoo, const double* X,
                      double* L) {
  const int nc = 1 << dim;
  double G[MAXD * 64];
  memset(L, 0, sizeof(double) * nn * nn);
  for (int g = 0; g < ng; ++g) {
    double det = point_geom(dim, nn, nc, Hcoo + (size_t)g * dim * nc, Hrs + (size_t)g * dim * nn, X, G);
    double c = w[g] * det;
    for (int a = 0; a < nn; ++a)
      for (int b = 0; b < nn; ++b) {
        double s = 0;
        for (int d = 0; d < dim; ++d) s += G[d * nn + a] * G[d * nn + b];
        L[a * nn + b] += c * s;
      }
  }
}

/* ---- node graph ---------------------------------------------------------------------------- */
static int cmp_i32(const void* a, const void* b) {
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return (x > y) - (x < y);
}

/* pass colidx == NULL to get rowptr only (rowptr[n_node] = nnz) */
int64_t orc_csr_pattern(int nn, int64_t n_elem, int64_t n_node, const int32_t* conn, int32_t* rowptr, int32_t* colidx) {
  int32_t* deg = (int32_t*)calloc((size_t)n_node + 1, sizeof(int32_t));
  for (int64_t i = 0; i < n_elem * nn; ++i) deg[conn[i] + 1]++;
  for (int64_t i = 0; i < n_node; ++i) deg[i + 1] += deg[i];
  int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_node);
  memcpy(cur, deg, sizeof(int32_t) * (size_t)n_node);
  int32_t* n2e = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_elem * nn));
  for (int64_t e = 0; e < n_elem; ++e)
    for (int a = 0; a < nn; ++a) n2e[cur[conn[e * nn + a]]++] = (int32_t)e;
  rowptr[0] = 0;
#pragma omp parallel
  {
    int32_t* buf = (int32_t*)malloc(sizeof(int32_t) * 4096);
    size_t cap = 4096;
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n_node; ++i) {
      size_t cnt = (size_t)(deg[i + 1] - deg[i]) * nn;
      if (cnt > cap) { cap = cnt; buf = (int32_t*)realloc(buf, sizeof(int32_t) * cap); }
      size_t m = 0;
      for (int32_t k = deg[i]; k < deg[i + 1]; ++k)
        for (int a = 0; a < nn; ++a) buf[m++] = conn[(int64_t)n2e[k] * nn + a];
      qsort(buf, m, sizeof(int32_t), cmp_i32);
      int32_t u = 0;
      for (size_t k = 0; k < m; ++k)
        if (k == 0 || buf[k] != buf[k - 1]) ++u;
      rowptr[i + 1] = u;
    }
    free(buf);
  }
  for (int64_t i = 0; i < n_node; ++i) rowptr[i + 1] += rowptr[i];
  int64_t nnz = rowptr[n_node];
  if (colidx) {
#pragma omp parallel
    {
      int32_t* buf = (int32_t*)malloc(sizeof(int32_t) * 4096);
      size_t cap = 4096;
#pragma omp for schedule(static)
      for (int64_t i = 0; i < n_node; ++i) {
        size_t cnt = (size_t)(deg[i + 1] - deg[i]) * nn;
        if (cnt > cap) { cap = cnt; buf = (int32_t*)realloc(buf, sizeof(int32_t) * cap); }
        size_t m = 0;
        for (int32_t k = deg[i]; k < deg[i + 1]; ++k)
          for (int a = 0; a < nn; ++a) buf[m++] = conn[(int64_t)n2e[k] * nn + a];
        qsort(buf, m, sizeof(int32_t), cmp_i32);
        int32_t o = rowptr[i];
        for (size_t k = 0; k < m; ++k)
          if (k == 0 || buf[k] != buf[k - 1]) colidx[o++] = buf[k];
      }
      free(buf);
    }
  }
  free(deg); free(cur); free(n2e);
  return nnz;
}

static int find_slot(const int32_t* colidx, int lo, int len, int col) {
  int l = 0, h = len;
  while (l < h) {
    int m = (l + h) >> 1;
    if (colidx[lo + m] < col) l = m + 1; else h = m;
  }
  return l;
}

/* scalar Laplace: A[free,free] += L_e ; Arhs[free,bc] += -L_e ; unit diagonal on bc.  Elements
 * [e0, e1) only (bounded samples for the CPU baseline). */
void orc_assemble_laplace(int dim, int nn, int64_t e0, int64_t e1, const int32_t* conn, const double* xyz, int ng,
                          const double* w, const double* Hrs, const double* Hcoo, const int32_t* rowptr,
                          const int32_t* colidx, const uint8_t* bc, int64_t n_node, double* A, double* Arhs) {
  const int nc = 1 << dim;
#pragma omp parallel for schedule(static)
  for (int64_t e = e0; e < e1; ++e) {
    double X[8 * MAXD], L[64 * 64];
    const int32_t* ce = conn + e * nn;
    for (int c = 0; c < nc; ++c)
      for (int x = 0; x < dim; ++x) X[c * dim + x] = xyz[(int64_t)ce[c] * dim + x];   /* dmplex.py:97-104 */
    orc_elem_laplace(dim, nn, ng, w, Hrs, Hcoo, X, L);
    for (int a = 0; a < nn; ++a) {
      int ra = ce[a];
      if (bc && bc[ra]) continue;
      int lo = rowptr[ra], len = rowptr[ra + 1] - lo;
      for (int b = 0; b < nn; ++b) {
        int slot = find_slot(colidx, lo, len, ce[b]);
        double v = L[a * nn + b];
        if (bc && bc[ce[b]]) {
          if (Arhs) {
#pragma omp atomic
            Arhs[lo + slot] -= v;
          }
        } else {
#pragma omp atomic
          A[lo + slot] += v;
        }
      }
    }
  }
  if (bc && e0 == 0) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_node; ++i)
      if (bc[i]) {
        int lo = rowptr[i], len = rowptr[i + 1] - lo;
        int s = find_slot(colidx, lo, len, (int)i);
        A[lo + s] = 1.0;
        if (Arhs) Arhs[lo + s] = 1.0;
      }
  }
}

/* KLE: K, Krhs (dim x dim blocks), Rw (dim x dw).  bc mask per velocity DOF. */
void orc_assemble_kle(int dim, int nn, int64_t e0, int64_t e1, const int32_t* conn, const double* xyz, int ngf,
                      const double* wf, const double* Hf, const double* Hrsf, const double* Hcoof, int ngr,
                      const double* wr, const double* Hr, const double* Hrsr, const double* Hcoor, double alpha_d,
                      double alpha_w, const int32_t* rowptr, const int32_t* colidx, const uint8_t* bc, int64_t n_node,
                      double* K, double* Krhs, double* Rw) {
  const int nc = 1 << dim, nd = dim * nn, dw = dim == 2 ? 1 : 3, nw = dw * nn;
#pragma omp parallel
  {
    double* Ke = (double*)malloc(sizeof(double) * nd * nd);
    double* Rwe = (double*)malloc(sizeof(double) * nd * nw);
#pragma omp for schedule(static)
    for (int64_t e = e0; e < e1; ++e) {
      double X[8 * MAXD];
      const int32_t* ce = conn + e * nn;
      for (int c = 0; c < nc; ++c)
        for (int x = 0; x < dim; ++x) X[c * dim + x] = xyz[(int64_t)ce[c] * dim + x];
      orc_elem_kle(dim, nn, ngf, wf, Hf, Hrsf, Hcoof, ngr, wr, Hr, Hrsr, Hcoor, X, alpha_d, alpha_w, Ke, Rw ? Rwe : 0, 0);
      for (int a = 0; a < nn; ++a) {
        int ra = ce[a];
        int lo = rowptr[ra], len = rowptr[ra + 1] - lo;
        for (int b = 0; b < nn; ++b) {
          int slot = find_slot(colidx, lo, len, ce[b]);
          for (int p = 0; p < dim; ++p) {
            if (bc && bc[(int64_t)ra * dim + p]) continue;                       /* base_problem.py:522-528 */
            for (int q = 0; q < dim; ++q) {
              double v = Ke[(size_t)(a * dim + p) * nd + b * dim + q];
              int64_t off = ((int64_t)lo * dim + (int64_t)p * len + slot) * dim + q;
              if (bc && bc[(int64_t)ce[b] * dim + q]) {
                if (Krhs) {
#pragma omp atomic
                  Krhs[off] -= v;                                               /* :531-533 */
                }
              } else {
#pragma omp atomic
                K[off] += v;                                                    /* :540-541 */
              }
            }
            if (Rw)
              for (int k = 0; k < dw; ++k) {
                int64_t off = ((int64_t)lo * dim + (int64_t)p * len + slot) * dw + k;
#pragma omp atomic
                Rw[off] += Rwe[(size_t)(a * dim + p) * nw + b * dw + k];        /* :546-547 */
              }
          }
        }
      }
    }
    free(Ke); free(Rwe);
  }
  if (bc && e0 == 0) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_node; ++i)
      for (int p = 0; p < dim; ++p)
        if (bc[i * dim + p]) {                                                  /* mat_generator.py:113-118 */
          int lo = rowptr[i], len = rowptr[i + 1] - lo;
          int s = find_slot(colidx, lo, len, (int)i);
          int64_t off = ((int64_t)lo * dim + (int64_t)p * len + s) * dim + p;
          K[off] = 1.0;
          if (Krhs) Krhs[off] = 1.0;
        }
  }
}

/* ---- SpMV + Jacobi PCG ------------------------------------------------------------------------ */
void orc_spmv(int64_t n_node, int br, int bc, const int32_t* rowptr, const int32_t* colidx, const double* val,
              const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_node; ++i) {
    int lo = rowptr[i], len = rowptr[i + 1] - lo;
    for (int p = 0; p < br; ++p) {
      const double* v = val + ((int64_t)lo * br + (int64_t)p * len) * bc;
      double s = 0;
      for (int k = 0; k < len; ++k)
        for (int q = 0; q < bc; ++q) s += v[k * bc + q] * x[(int64_t)colidx[lo + k] * bc + q];
      y[i * br + p] = s;
    }
  }
}

static double dotp(int64_t n, const double* a, const double* b) {
  double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* norm_type: 0 preconditioned, 1 unpreconditioned, 2 natural.  fixed_iters > 0 disables the exit test.
 * returns iterations; *rnorm_out last norm */
int orc_pcg(int64_t n_node, int bs, const int32_t* rowptr, const int32_t* colidx, const double* val, const double* b,
            double* x, double rtol, double atol, int maxit, int norm_type, int fixed_iters, double* rnorm_out) {
  const int64_t n = n_node * bs;
  double* r = (double*)malloc(sizeof(double) * n);
  double* z = (double*)malloc(sizeof(double) * n);
  double* p = (double*)malloc(sizeof(double) * n);
  double* Ap = (double*)malloc(sizeof(double) * n);
  double* dinv = (double*)malloc(sizeof(double) * n);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_node; ++i) {
    int lo = rowptr[i], len = rowptr[i + 1] - lo;
    int s = find_slot(colidx, lo, len, (int)i);
    for (int q = 0; q < bs; ++q) dinv[i * bs + q] = 1.0 / val[((int64_t)lo * bs + (int64_t)q * len + s) * bs + q];
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) { x[i] = 0; r[i] = b[i]; z[i] = dinv[i] * r[i]; p[i] = z[i]; }
  double rz = dotp(n, r, z);
  double rn = norm_type == 0 ? sqrt(dotp(n, z, z)) : norm_type == 1 ? sqrt(dotp(n, r, r)) : sqrt(fabs(rz));
  double ttol = fmax(rtol * rn, atol);
  int it = 0;
  const int lim = fixed_iters > 0 ? fixed_iters : maxit;
  if (fixed_iters > 0 || rn > ttol) {
    for (it = 1; it <= lim; ++it) {
      orc_spmv(n_node, bs, bs, rowptr, colidx, val, p, Ap);
      double alpha = rz / dotp(n, p, Ap);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) { x[i] += alpha * p[i]; r[i] -= alpha * Ap[i]; z[i] = dinv[i] * r[i]; }
      double rzn = dotp(n, r, z);
      rn = norm_type == 0 ? sqrt(dotp(n, z, z)) : norm_type == 1 ? sqrt(dotp(n, r, r)) : sqrt(fabs(rzn));
      if (fixed_iters <= 0 && rn <= ttol) break;
      double beta = rzn / rz;
      rz = rzn;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    }
    if (it > lim) it = lim;
  }
  if (rnorm_out) *rnorm_out = rn;
  free(r); free(z); free(p); free(Ap); free(dinv);
  return it;
}
