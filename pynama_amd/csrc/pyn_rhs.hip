// Compact "imposed-column" matrices: Krhs / Krhsfs / Arhs of an assembly hold -K_e[free, bc] (+ the unit diagonal of the imposed DOFs)
// and nothing else, i.e. they are zero outside the node rows that have an imposed node in their neighbourhood.  The reference
// preallocates them accordingly (src/matrices/mat_generator.py:42-58, 91: `drhs_nnz` counts the Dirichlet columns of a row); here a
// matrix created by pyn_mat_create_rhs stores exactly those node rows, each with the graph's full column list (the rows keep the
// block-CSR layout of include/pynama_hip.h, so every kernel that can address a row can fill it).  At 128^3 / 3 DOFs per node that is
// 0.37 GB instead of 4.1 GB, and Krhs v in every solveKLE (src/cases/base_problem.py:481) streams the boundary layer only.
#include <hipcub/hipcub.hpp>
#include <hipcub/iterator/counting_input_iterator.hpp>

#include "pyn_internal.h"

namespace {

// f[i] = 1 iff node row i has an imposed node among its columns (itself included); len[i] = f ? blocks of the row : 0
__global__ void rhs_flag_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, const uint8_t* __restrict__ mask,
                                int ndof, int64_t n_rows, int32_t* __restrict__ flag, int32_t* __restrict__ len) {
  const int lane = threadIdx.x & 63;
  const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = w; i <= n_rows; i += nw) {
    if (i == n_rows) {
      if (lane == 0) flag[i] = len[i] = 0;
      continue;
    }
    const int lo = rowptr[i], ln = rowptr[i + 1] - lo;
    int f = 0;
    for (int k = lane; k < ln && mask; k += 64) {
      const int64_t j = colidx[lo + k];
      for (int q = 0; q < ndof; ++q) f |= mask[j * ndof + q];
    }
    f = __any(f != 0);
    if (lane == 0) {
      flag[i] = f ? 1 : 0;
      len[i] = f ? ln : 0;
    }
  }
}

__global__ void rhs_crow_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ scan, int64_t n_rows, int32_t* __restrict__ crow) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_rows) crow[i] = flag[i] ? scan[i] : -1;
}

__global__ void rhs_cptr_kernel(const int32_t* __restrict__ rsel, const int32_t* __restrict__ scan, int64_t nr, int64_t n_rows,
                                int32_t* __restrict__ cptr) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nr) cptr[k] = scan[rsel[k]];
  if (k == nr) cptr[k] = scan[n_rows];
}

__global__ void rhs_expand_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rsel, const int32_t* __restrict__ cptr,
                                  const double* __restrict__ val, int64_t nr, int bb, double* __restrict__ full) {
  const int lane = threadIdx.x & 63;
  const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t k = w; k < nr; k += nw) {
    const int64_t i = rsel[k];
    const int64_t src = (int64_t)cptr[k] * bb, dst = (int64_t)rowptr[i] * bb;
    const int n = (cptr[k + 1] - cptr[k]) * bb;
    for (int t = lane; t < n; t += 64) full[dst + t] = val[src + t];
  }
}

__global__ void bc_elem_flag_kernel(const int32_t* __restrict__ conn, int nn, int64_t n_elem, const uint8_t* __restrict__ mask, int ndof,
                                    uint8_t* __restrict__ flag) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_elem) return;
  int f = 0;
  for (int a = 0; a < nn; ++a) {
    const int64_t j = conn[e * nn + a];
    for (int q = 0; q < ndof; ++q) f |= mask[j * ndof + q];
  }
  flag[e] = f ? 1 : 0;
}

}  // namespace

void pyn_rhs_release(DMat& M) {
  (void)hipFree(M.c_crow);
  (void)hipFree(M.c_rsel);
  (void)hipFree(M.c_cptr);
  M.c_crow = M.c_rsel = M.c_cptr = nullptr;
  M.c_nr = M.c_nnzb = 0;
  M.c_stamp = -1;
}

int64_t pyn_mat_blocks(const pyn_ctx* c, const DMat& M) { return M.rhs_compact ? M.c_nnzb : c->nnzb; }

// relayout: the caller is about to ASSEMBLE into the matrix -- its row selection must be that of the current Dirichlet set.  Every
// other use (products, host insertion, read-back) keeps the layout the values were assembled for, whatever pyn_bc_set did since.
int pyn_rhs_ensure(pyn_ctx* c, DMat& M, bool relayout) {
  if (!M.rhs_compact || (M.c_crow && (!relayout || M.c_stamp == c->bc_stamp))) return PYN_OK;
  hipStream_t s = c->stream;
  const int64_t n = c->n_owned;
  PYN_HIP(hipStreamSynchronize(s));   // no kernel in flight may still read the old arrays
  (void)hipFree(M.val);
  M.val = nullptr;
  pyn_rhs_release(M);
  DevTmp tflag, tlen, tscan, tmp, tnr;
  PYN_HIP(tflag.alloc((n + 1) * sizeof(int32_t)));
  PYN_HIP(tlen.alloc((n + 1) * sizeof(int32_t)));
  PYN_HIP(tscan.alloc((n + 1) * sizeof(int32_t)));
  PYN_HIP(tnr.alloc(sizeof(int64_t)));
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 1 + 3) / 4, 16384));
  rhs_flag_kernel<<<grid, 256, 0, s>>>(c->d_rowptr, c->d_colidx, c->d_bcmask, c->bc_ndof, n, tflag.as<int32_t>(), tlen.as<int32_t>());
  size_t tb = 0;
  PYN_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, tlen.as<int32_t>(), tscan.as<int32_t>(), (int)(n + 1), s));
  PYN_HIP(tmp.alloc(tb));
  PYN_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, tlen.as<int32_t>(), tscan.as<int32_t>(), (int)(n + 1), s));
  PYN_HIP(hipMalloc((void**)&M.c_crow, std::max<int64_t>(n, 1) * sizeof(int32_t)));
  PYN_HIP(hipMalloc((void**)&M.c_rsel, std::max<int64_t>(n, 1) * sizeof(int32_t)));   // upper bound; trimmed below
  rhs_crow_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(tflag.as<int32_t>(), tscan.as<int32_t>(), n, M.c_crow);
  hipcub::CountingInputIterator<int32_t> ids(0);
  tb = 0;
  PYN_HIP(hipcub::DeviceSelect::Flagged(nullptr, tb, ids, tflag.as<int32_t>(), M.c_rsel, tnr.as<int64_t>(), (int)n, s));
  PYN_HIP(tmp.alloc(tb));
  PYN_HIP(hipcub::DeviceSelect::Flagged(tmp.p, tb, ids, tflag.as<int32_t>(), M.c_rsel, tnr.as<int64_t>(), (int)n, s));
  int64_t nr = 0;
  int32_t total = 0;
  PYN_HIP(hipMemcpyAsync(&nr, tnr.p, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  PYN_HIP(hipMemcpyAsync(&total, tscan.as<int32_t>() + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PYN_HIP(hipStreamSynchronize(s));
  PYN_HIP(hipMalloc((void**)&M.c_cptr, (nr + 1) * sizeof(int32_t)));
  rhs_cptr_kernel<<<(int)((nr + 1 + 255) / 256), 256, 0, s>>>(M.c_rsel, tscan.as<int32_t>(), nr, n, M.c_cptr);
  M.c_nr = nr;
  M.c_nnzb = total;
  const size_t bytes = (size_t)std::max<int64_t>(total, 1) * M.br * M.bc * sizeof(double);
  PYN_HIP(hipMalloc((void**)&M.val, bytes));
  PYN_HIP(hipMemsetAsync(M.val, 0, bytes, s));
  PYN_HIP(hipStreamSynchronize(s));   // the scratch arrays go out of scope
  M.touch();
  M.rhs_clean = PYN_RHS_ANY;
  M.c_stamp = c->bc_stamp;
  return PYN_OK;
}

int pyn_rhs_expand(pyn_ctx* c, const DMat& M, double* full) {
  const int bb = M.br * M.bc;
  PYN_HIP(hipMemsetAsync(full, 0, (size_t)c->nnzb * bb * sizeof(double), c->stream));
  if (M.c_nr > 0) {
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((M.c_nr + 3) / 4, 16384));
    rhs_expand_kernel<<<grid, 256, 0, c->stream>>>(c->d_rowptr, M.c_rsel, M.c_cptr, M.val, M.c_nr, bb, full);
    PYN_HIP(hipGetLastError());
  }
  return PYN_OK;
}

// elements that hold an imposed node: the only ones whose K_e[free, bc] is not empty
int pyn_bc_elements(pyn_ctx* c) {
  if (c->esel_stamp == c->bc_stamp && c->d_esel) return PYN_OK;
  hipStream_t s = c->stream;
  (void)hipFree(c->d_esel);
  c->d_esel = nullptr;
  c->n_esel = 0;
  if (!c->d_bcmask) {
    PYN_HIP(hipMalloc((void**)&c->d_esel, sizeof(int32_t)));
    c->esel_stamp = c->bc_stamp;
    return PYN_OK;
  }
  DevTmp tflag, tmp, tn;
  PYN_HIP(tflag.alloc((size_t)c->n_elem));
  PYN_HIP(tn.alloc(sizeof(int64_t)));
  bc_elem_flag_kernel<<<(int)((c->n_elem + 255) / 256), 256, 0, s>>>(c->d_conn, c->nn, c->n_elem, c->d_bcmask, c->bc_ndof, tflag.as<uint8_t>());
  PYN_HIP(hipMalloc((void**)&c->d_esel, (size_t)c->n_elem * sizeof(int32_t)));
  hipcub::CountingInputIterator<int32_t> ids(0);
  size_t tb = 0;
  PYN_HIP(hipcub::DeviceSelect::Flagged(nullptr, tb, ids, tflag.as<uint8_t>(), c->d_esel, tn.as<int64_t>(), (int)c->n_elem, s));
  PYN_HIP(tmp.alloc(tb));
  PYN_HIP(hipcub::DeviceSelect::Flagged(tmp.p, tb, ids, tflag.as<uint8_t>(), c->d_esel, tn.as<int64_t>(), (int)c->n_elem, s));
  PYN_HIP(hipMemcpyAsync(&c->n_esel, tn.p, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  PYN_HIP(hipStreamSynchronize(s));
  c->esel_stamp = c->bc_stamp;
  return PYN_OK;
}

extern "C" int pyn_mat_create_rhs(pyn_ctx* c, int br, int bc, int* mat_id) {
  PYN_CHECK(c && mat_id, "NULL argument");
  PYN_CHECK(c->d_rowptr, "pyn_csr_symbolic first");
  PYN_CHECK(br >= 1 && br <= 6 && bc >= 1 && bc <= 6, "block shape out of range");
  PYN_HIP(hipSetDevice(c->device));
  DMat m;
  m.br = br;
  m.bc = bc;
  m.rhs_compact = true;
  m.live = true;
  c->mats.push_back(m);
  *mat_id = (int)c->mats.size() - 1;
  return pyn_rhs_ensure(c, c->mats.back(), true);   // rows of the CURRENT Dirichlet set (laid out again by an assembly under another one)
}

extern "C" int pyn_mat_stored_blocks(pyn_ctx* c, int mat_id, int64_t* blocks, int64_t* node_rows) {
  PYN_TRY(pyn_check_mat(c, mat_id, "pyn_mat_stored_blocks"));
  const DMat& m = c->mats[mat_id];
  if (blocks) *blocks = pyn_mat_blocks(c, m);
  if (node_rows) *node_rows = m.rhs_compact ? m.c_nr : c->n_owned;
  return PYN_OK;
}
