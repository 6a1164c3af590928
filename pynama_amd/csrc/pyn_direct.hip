// Direct solve for SMALL systems: dense LU with partial pivoting on the device.
//
// The reference's hard-wired default is `-ksp_type preonly -pc_type lu` (src/solver/ksp_solver.py:13-16, makefile:7): PETSc
// factors K once at KSPSetUp and every solveKLE is two triangular solves.  A sparse direct solver is outside this build's hot
// path; what the reference's own tests and cases exercise with that default are systems of 10^2..10^3 unknowns (src/tests/
// test_solver.py: 882, 882 and 1,029 DOFs), and for those a dense factorisation IS a direct solve: block CSR -> dense n x n,
// blocked right-looking LU with partial pivoting (panels of 64 columns: panel factorisation, row interchanges, triangular solve, tile update), factors cached
// per matrix version, forward / backward substitution in blocks of 64 unknowns (one launch per block).  n <= PYN_DIRECT_MAX_N (8,192: 512 MB of factors); larger
// systems take the Krylov substitute of KspSolver (pynama_amd/solver/ksp_solver.py).  One rank only.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "pyn_internal.h"

namespace {

constexpr int64_t PYN_DIRECT_MAX_N = 8192;

__global__ void dense_from_bcsr_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, const double* __restrict__ val,
                                       int64_t n_nodes, int b, double* __restrict__ D, int64_t n) {
  // one thread per stored scalar entry of node row i: layout val[(rowptr[i]*b + p*len + k)*b + q]
  for (int64_t i = blockIdx.x; i < n_nodes; i += gridDim.x) {
    const int lo = rowptr[i], len = rowptr[i + 1] - lo;
    for (int e = threadIdx.x; e < len * b * b; e += blockDim.x) {
      const int p = e / (len * b), rem = e - p * len * b, k = rem / b, q = rem - k * b;
      D[(i * b + p) * n + (int64_t)colidx[lo + k] * b + q] = val[((int64_t)lo * b) * b + e];
    }
  }
}

// ---- blocked right-looking LU, panels of LU_NB columns: (1) inside a panel the elimination goes column by column with partial
// pivoting (pivot search in one workgroup, whole-row interchange, multipliers, rank-1 update of the panel's remaining columns: four
// small launches per column -- a one-workgroup panel kernel was tried and is slower: one CU streams the panel 64 times), (2) U12 =
// L11^-1 A12 (one thread per column, L11 through scalar loads), (3) A22 -= L21 U12 (64 x 64 tiles, 4 x 4 per thread, K = LU_NB
// through LDS).  The trailing matrix is touched once per panel instead of once per column.
constexpr int LU_NB = 64;

// pivot of column k: row of the largest |D[i][k]|, i >= k (one workgroup)
__global__ void __launch_bounds__(256) lu_pivot_kernel(const double* __restrict__ D, int64_t n, int k, int* __restrict__ piv, int* __restrict__ flag) {
  __shared__ double sv[256];
  __shared__ int si[256];
  double best = -1.0;
  int bi = k;
  for (int i = k + threadIdx.x; i < n; i += 256) {
    const double a = fabs(D[(int64_t)i * n + k]);
    if (a > best) {
      best = a;
      bi = i;
    }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s && (sv[threadIdx.x + s] > sv[threadIdx.x] ||
                            (sv[threadIdx.x + s] == sv[threadIdx.x] && si[threadIdx.x + s] < si[threadIdx.x]))) {
      sv[threadIdx.x] = sv[threadIdx.x + s];
      si[threadIdx.x] = si[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    piv[k] = si[0];
    if (!(sv[0] > 0.0)) *flag = k + 1;   // singular (or NaN) column
  }
}

__global__ void lu_swap_kernel(double* __restrict__ D, int64_t n, int k, const int* __restrict__ piv) {
  const int p = piv[k];
  if (p == k) return;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    const double a = D[(int64_t)k * n + j];
    D[(int64_t)k * n + j] = D[(int64_t)p * n + j];
    D[(int64_t)p * n + j] = a;
  }
}

// multipliers of column k, stored in place
__global__ void lu_scale_kernel(double* __restrict__ D, int64_t n, int k) {
  const double inv = 1.0 / D[(int64_t)k * n + k];
  for (int64_t i = k + 1 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) D[i * n + k] *= inv;
}

// rank-1 update inside the panel: D[i][j] -= l_ik D[k][j],  i > k,  k < j < kend  (16 x 16 tiles)
__global__ void __launch_bounds__(256) lu_update_kernel(double* __restrict__ D, int64_t n, int k, int kend) {
  const int64_t j = k + 1 + (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
  const int64_t i = k + 1 + (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= n || j >= kend) return;
  D[i * n + j] = fma(-D[i * n + k], D[(int64_t)k * n + j], D[i * n + j]);
}

// U12 = L11^-1 A12: one thread per column right of the panel, its nb entries in registers; L11 (unit lower, final after the panel
// kernel, disjoint from the columns written here) is the same for every thread: uniform addresses, i.e. scalar loads
__global__ void __launch_bounds__(256) lu_trsm_kernel(double* __restrict__ D, const double* __restrict__ L11, int64_t n, int kb, int nb) {
  const int64_t c = kb + nb + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  double u[LU_NB];
#pragma unroll
  for (int j = 0; j < LU_NB; ++j) u[j] = j < nb ? D[(int64_t)(kb + j) * n + c] : 0.0;
#pragma unroll
  for (int i = 1; i < LU_NB; ++i) {
    if (i < nb) {
      const double* __restrict__ Li = L11 + (int64_t)i * n;
      double a = u[i];
#pragma unroll
      for (int j = 0; j < i; ++j) a = fma(-Li[j], u[j], a);
      u[i] = a;
    }
  }
#pragma unroll
  for (int j = 1; j < LU_NB; ++j)
    if (j < nb) D[(int64_t)(kb + j) * n + c] = u[j];
}

// A22 -= L21 U12: 64 x 64 tile per workgroup, 4 x 4 per thread
__global__ void __launch_bounds__(256) lu_gemm_kernel(double* __restrict__ D, int64_t n, int kb, int nb) {
  __shared__ double As[64][LU_NB + 1];   // L21 tile [row][k]
  __shared__ double Bs[LU_NB][64];       // U12 tile [k][col]
  const int64_t i0 = kb + nb + (int64_t)blockIdx.y * 64, c0 = kb + nb + (int64_t)blockIdx.x * 64;
  const int t = threadIdx.x;
  for (int e = t; e < 64 * LU_NB; e += 256) {
    const int r = e / LU_NB, k = e % LU_NB;
    As[r][k] = (i0 + r < n && k < nb) ? D[(i0 + r) * n + kb + k] : 0.0;
  }
  for (int e = t; e < LU_NB * 64; e += 256) {
    const int k = e / 64, cc = e % 64;
    Bs[k][cc] = (c0 + cc < n && k < nb) ? D[(int64_t)(kb + k) * n + c0 + cc] : 0.0;
  }
  __syncthreads();
  const int tx = t & 15, ty = t >> 4;
  double acc[4][4] = {};
#pragma unroll 8
  for (int k = 0; k < LU_NB; ++k) {
    double a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = As[ty * 4 + r][k];
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = Bs[k][tx + 16 * q];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[r][q] = fma(a[r], b[q], acc[r][q]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t i = i0 + ty * 4 + r, c = c0 + tx + 16 * q;
      if (i < n && c < n) D[i * n + c] -= acc[r][q];
    }
}

// the interchanges of the factorisation as one gather: (P b)[i] = b[perm[i]]  (once per factorisation, one thread)
__global__ void lu_perm_kernel(const int* __restrict__ piv, int n, int* __restrict__ perm) {
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int k = 0; k < n; ++k) {
    const int p = piv[k];
    if (p != k) {
      const int a = perm[k];
      perm[k] = perm[p];
      perm[p] = a;
    }
  }
}

// value of lane k (compile-time constant after unrolling) in every lane: two v_readlane_b32
__device__ __forceinline__ double lane_bcast(double v, int k) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}

// x = U^-1 L^-1 P b in blocks of 64 unknowns, one launch per block and direction (2 ceil(n / 64) launches; the launch boundary is
// the grid-wide barrier between "block solved" and "block eliminated from the other rows").  Every workgroup first solves the 64 x 64
// diagonal block redundantly in its wave 0 -- the block's rows in registers (lane = row), 64 unrolled steps of readlane + fma, no
// memory traffic between the steps -- then the grid subtracts the block's columns from the remaining rows: 16 lanes per row read
// its 64 contiguous doubles (four each), a DPP reduction finishes the dot product.  n = 1,029: 2.0 ms (column-by-column sweep in one
// workgroup) -> 0.2 ms per solve.
__global__ void lu_gather_kernel(const double* __restrict__ b, const int* __restrict__ perm, int64_t n, double* __restrict__ w) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) w[i] = b[perm[i]];
}

template <bool FWD>
__global__ void __launch_bounds__(256) lu_block_step_kernel(const double* __restrict__ D, int64_t n, int64_t kb, double* __restrict__ w,
                                                            double* __restrict__ out) {
  __shared__ double ws[64];
  const int t = threadIdx.x, lane = t & 63;
  const int nb = (int)min((int64_t)64, n - kb);
  if (t < 64) {
    const int64_t row = min(kb + lane, n - 1);
    double R[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) R[j] = D[row * n + min(kb + j, n - 1)];
    double wv = w[row];
    if (FWD) {
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        const double wk = lane_bcast(wv, k);
        if (lane > k && k < nb) wv = fma(-R[k], wk, wv);
      }
    } else {
#pragma unroll
      for (int k = 63; k >= 0; --k) {
        if (lane == k && k < nb) wv = wv / R[k];
        const double wk = lane_bcast(wv, k);
        if (lane < k && k < nb) wv = fma(-R[k], wk, wv);
      }
    }
    ws[lane] = lane < nb ? wv : 0.0;
  }
  __syncthreads();
  // rows still to be updated: below the block (forward) / above it (backward); 16 lanes per row
  const int64_t r0 = FWD ? kb + 64 : 0, r1 = FWD ? n : kb;
  const int sub = t >> 4, q = t & 15;
  const double s0 = ws[4 * q], s1 = ws[4 * q + 1], s2 = ws[4 * q + 2], s3 = ws[4 * q + 3];
  for (int64_t i = r0 + (int64_t)blockIdx.x * 16 + sub; i < r1; i += (int64_t)gridDim.x * 16) {
    const double* __restrict__ r = D + i * n + kb + 4 * q;
    double a = 0.0;
    if (4 * q + 3 < nb) a = fma(r[3], s3, fma(r[2], s2, fma(r[1], s1, r[0] * s0)));
    else
      for (int j = 0; j < 4; ++j)
        if (4 * q + j < nb) a = fma(r[j], ws[4 * q + j], a);
    a += __shfl_xor(a, 8, 16);
    a += __shfl_xor(a, 4, 16);
    a += __shfl_xor(a, 2, 16);
    a += __shfl_xor(a, 1, 16);
    if (q == 0) w[i] -= a;
  }
  // the solved block goes to the OTHER array: workgroups of this launch that start late must still find the unsolved entries in w
  // (w[kb ..] is read-only here, the update touches disjoint rows, each once)
  if (blockIdx.x == 0 && t < nb) out[kb + t] = ws[t];
}

// out[0] = ||b - w||^2, out[1] = ||b||^2 (one workgroup)
__global__ void __launch_bounds__(1024) direct_resid_kernel(const double* __restrict__ w, const double* __restrict__ b, int64_t n, double* __restrict__ out) {
  __shared__ double sr[1024], sb[1024];
  double r = 0.0, q = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double d = b[i] - w[i];
    r = fma(d, d, r);
    q = fma(b[i], b[i], q);
  }
  sr[threadIdx.x] = r;
  sb[threadIdx.x] = q;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      sr[threadIdx.x] += sr[threadIdx.x + s];
      sb[threadIdx.x] += sb[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sr[0];
    out[1] = sb[0];
  }
}

}  // namespace

// factors of A, cached in the matrix until its values change
static int direct_factor(pyn_ctx* c, DMat& A) {
  const int64_t n = c->n_owned * A.br;
  if (A.lu_valid && A.lu_n == n) return PYN_OK;
  if (A.lu_n != n) A.release_lu();
  if (!A.lu) PYN_HIP(hipMalloc((void**)&A.lu, (size_t)n * n * sizeof(double)));
  if (!A.lu_piv) PYN_HIP(hipMalloc((void**)&A.lu_piv, (size_t)(2 * n + 1) * sizeof(int)));
  A.lu_n = n;
  hipStream_t s = c->stream;
  PYN_HIP(hipMemsetAsync(A.lu, 0, (size_t)n * n * sizeof(double), s));
  int* flag = A.lu_piv + n;
  PYN_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
  dense_from_bcsr_kernel<<<(int)std::min<int64_t>(c->n_owned, 4096), 256, 0, s>>>(c->d_rowptr, c->d_colidx, A.val, c->n_owned, A.br, A.lu, n);
  for (int kb = 0; kb < (int)n; kb += LU_NB) {
    const int nb = (int)std::min<int64_t>(LU_NB, n - kb), kend = kb + nb;
    for (int k = kb; k < kend; ++k) {
      lu_pivot_kernel<<<1, 256, 0, s>>>(A.lu, n, k, A.lu_piv, flag);
      lu_swap_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(A.lu, n, k, A.lu_piv);
      const int m = (int)(n - k - 1), w = kend - k - 1;
      if (m > 0) lu_scale_kernel<<<(m + 255) / 256, 256, 0, s>>>(A.lu, n, k);
      if (m > 0 && w > 0) lu_update_kernel<<<dim3((w + 15) / 16, (m + 15) / 16), 256, 0, s>>>(A.lu, n, k, kend);
    }
    const int64_t m2 = n - kend;
    if (m2 > 0) {
      lu_trsm_kernel<<<(int)((m2 + 255) / 256), 256, 0, s>>>(A.lu, A.lu + (int64_t)kb * n + kb, n, kb, nb);
      const int g = (int)((m2 + 63) / 64);
      lu_gemm_kernel<<<dim3(g, g), 256, 0, s>>>(A.lu, n, kb, nb);
    }
  }
  lu_perm_kernel<<<1, 1, 0, s>>>(A.lu_piv, (int)n, A.lu_piv + n + 1);
  int h = 0;
  PYN_HIP(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, s));
  PYN_HIP(hipStreamSynchronize(s));
  PYN_HIP(hipGetLastError());
  PYN_CHECK(h == 0, "direct solve: zero pivot in column %d of %lld (the matrix is singular)", h - 1, (long long)n);
  A.lu_valid = true;
  return PYN_OK;
}

extern "C" int pyn_direct_max_rows(void) { return (int)PYN_DIRECT_MAX_N; }

extern "C" int pyn_solve_direct(pyn_ctx* c, int mat_id, int bv, int xv, pyn_solve_info* info) {
  PYN_TRY(pyn_check_mat(c, mat_id, "pyn_solve_direct"));
  PYN_TRY(pyn_check_vec(c, bv, "pyn_solve_direct b"));
  PYN_TRY(pyn_check_vec(c, xv, "pyn_solve_direct x"));
  PYN_CHECK(info, "NULL argument");
  PYN_CHECK(bv != xv, "b and x must differ");
  DMat& A = c->mats[mat_id];
  PYN_CHECK(A.br == A.bc, "matrix must be square");
  PYN_CHECK(!A.rhs_compact, "a compact imposed-column matrix (pyn_mat_create_rhs) is a right-hand-side operator, not a system matrix");
  PYN_CHECK(c->vecs[bv].bs == A.br && c->vecs[xv].bs == A.br, "vector block size mismatch");
  PYN_CHECK(c->nranks == 1 && c->n_ghost == 0, "the dense direct solve runs on one rank");
  const int64_t n = c->n_owned * A.br;
  PYN_CHECK(n <= PYN_DIRECT_MAX_N, "direct solve: %lld rows exceed the dense limit of %lld (use the Krylov solvers)", (long long)n,
            (long long)PYN_DIRECT_MAX_N);
  PYN_HIP(hipSetDevice(c->device));
  memset(info, 0, sizeof(*info));
  PYN_HIP(hipEventRecord(c->ev0, c->stream));
  PYN_TRY(direct_factor(c, A));
  PYN_TRY(pyn_ensure_work(c, (size_t)2 * n * sizeof(double)));
  double* b = c->vecs[bv].d;
  double* x = c->vecs[xv].d;
  hipStream_t s = c->stream;
  double* z = c->d_work;
  lu_gather_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(b, A.lu_piv + n + 1, n, x);     // x = P b
  for (int64_t kb = 0; kb < n; kb += 64) {       // forward: blocks of x solved into z, the rows below updated in x
    const int64_t rows = std::max<int64_t>(n - kb - 64, 0);
    lu_block_step_kernel<true><<<(int)std::max<int64_t>(1, std::min<int64_t>((rows + 15) / 16, 1024)), 256, 0, s>>>(A.lu, n, kb, x, z);
  }
  for (int64_t kb = ((n - 1) / 64) * 64; kb >= 0; kb -= 64)   // backward: blocks of z solved into x, the rows above updated in z
    lu_block_step_kernel<false><<<(int)std::max<int64_t>(1, std::min<int64_t>((kb + 15) / 16, 1024)), 256, 0, s>>>(A.lu, n, kb, z, x);
  PYN_HIP(hipEventRecord(c->ev1, c->stream));
  // true residual through the sparse matrix
  double* w = c->d_work + n;
  PYN_TRY(pyn_spmv_raw(c, A, x, w));
  direct_resid_kernel<<<1, 1024, 0, c->stream>>>(w, b, n, c->d_scal);
  PYN_HIP(hipMemcpyAsync(c->h_scal, c->d_scal, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  PYN_HIP(hipGetLastError());
  const double rr = c->h_scal[0], bb = c->h_scal[1];
  float ms = 0;
  PYN_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  info->solve_ms = ms;
  info->iters = 1;                       // PETSc reports one "iteration" for preonly
  info->true_resid = bb > 0 ? sqrt(rr / bb) : sqrt(rr);
  info->rnorm = sqrt(rr);
  info->rnorm0 = sqrt(bb);
  info->reason = (info->true_resid == info->true_resid) ? PYN_CONVERGED_ITS : PYN_DIVERGED_NANORINF;
  c->timers[PYN_T_SOLVE] = ms;
  return PYN_OK;
}
