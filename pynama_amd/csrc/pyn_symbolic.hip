// Symbolic phase: node adjacency graph -> device CSR (rows = owned nodes).
// Takes over DMPlexDom.getMatIndices (src/domain/dmplex.py:305-333: per-row Python set
// construction) and the nnz preallocation of Mat.createEmptyKLEMats (mat_generator.py:32-99).
//
// Method: every element emits its nn*nn (row, col) node pairs as 64-bit keys, a device radix
// sort + unique (hipCUB, plumbing only) yields the sorted pattern for any element order /
// any mesh; rows owned by other ranks are dropped.
#include <hipcub/hipcub.hpp>

#include "pyn_internal.h"

__global__ void emit_pairs_kernel(const int32_t* __restrict__ conn, int64_t n_elem, int nn, int64_t n_owned,
                                  unsigned long long* __restrict__ keys) {
  int64_t total = n_elem * nn * nn;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t e = t / (nn * nn);
    int ab = (int)(t - e * nn * nn);
    int a = ab / nn, b = ab - a * nn;
    unsigned long long r = (unsigned long long)conn[e * nn + a];
    unsigned long long cidx = (unsigned long long)conn[e * nn + b];
    keys[t] = r < (unsigned long long)n_owned ? ((r << 32) | cidx) : ~0ull;
  }
}

__global__ void split_keys_kernel(const unsigned long long* __restrict__ keys, int64_t n, int32_t* __restrict__ colidx,
                                  int32_t* __restrict__ rowcnt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long k = keys[i];
    colidx[i] = (int32_t)(k & 0xffffffffull);
    atomicAdd(&rowcnt[(int32_t)(k >> 32)], 1);
  }
}

// the patch plans, the SELL structures and all matrices are tied to the graph
static int pyn_symbolic_reset_dependents(pyn_ctx* c) {
  PYN_TRY(pyn_patch_plan_set_kind(c, 0, 0, nullptr, nullptr));
  PYN_TRY(pyn_patch_plan_set_kind(c, 1, 0, nullptr, nullptr));
  for (auto& m : c->mats) {
    (void)hipFree(m.val);
    (void)hipFree(m.sell_val);
    (void)hipFree(m.dinv);
    m.release_lu();
    pyn_rhs_release(m);
  }
  c->mats.clear();
  c->esel_stamp = -1;
  pyn_sell_drop_structure(c);
  c->lat.std_ok = -1;  // closed-form row offsets are re-verified against the new graph
  c->plan_unfit[0] = c->plan_unfit[1] = false;
  return PYN_OK;
}

extern "C" int pyn_csr_symbolic(pyn_ctx* c) {
  PYN_CHECK(c, "ctx is NULL");
  PYN_CHECK(c->n_elem > 0, "pyn_mesh_set first");
  PYN_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  PYN_HIP(hipEventRecord(c->ev0, s));
  bool arithmetic = false;   // structured topology: the graph in closed form, no sort
  PYN_TRY(pyn_lattice_symbolic(c, &arithmetic));
  if (!arithmetic) PYN_TRY(pyn_ho3_symbolic(c, &arithmetic));
  if (arithmetic) {
    PYN_HIP(hipEventRecord(c->ev1, s));
    PYN_HIP(hipStreamSynchronize(s));
    float ms0 = 0;
    PYN_HIP(hipEventElapsedTime(&ms0, c->ev0, c->ev1));
    c->timers[PYN_T_SYMBOLIC] = ms0;
    return pyn_symbolic_reset_dependents(c);
  }
  const int64_t total = c->n_elem * c->nn * c->nn;
  DevTmp tk0, tk1, tn, tmp, tcnt;
  PYN_HIP(tk0.alloc(total * sizeof(unsigned long long)));
  PYN_HIP(tk1.alloc(total * sizeof(unsigned long long)));
  PYN_HIP(tn.alloc(sizeof(int64_t)));
  unsigned long long *k0 = tk0.as<unsigned long long>(), *k1 = tk1.as<unsigned long long>();
  int64_t* d_nuniq = tn.as<int64_t>();
  int grid = (int)std::min<int64_t>((total + 255) / 256, 65536);
  emit_pairs_kernel<<<grid, 256, 0, s>>>(c->d_conn, c->n_elem, c->nn, c->n_owned, k0);
  // invalid keys (rows of other ranks) are all-ones: sort over the full width keeps them last
  size_t tb = 0;
  PYN_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, k0, k1, total, 0, 64, s));
  PYN_HIP(tmp.alloc(tb));
  PYN_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, k0, k1, total, 0, 64, s));
  tb = 0;
  PYN_HIP(hipcub::DeviceSelect::Unique(nullptr, tb, k1, k0, d_nuniq, total, s));
  PYN_HIP(hipStreamSynchronize(s));
  PYN_HIP(tmp.alloc(tb));
  PYN_HIP(hipcub::DeviceSelect::Unique(tmp.p, tb, k1, k0, d_nuniq, total, s));
  int64_t nuniq = 0;
  unsigned long long last = 0;
  PYN_HIP(hipMemcpyAsync(&nuniq, d_nuniq, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  PYN_HIP(hipStreamSynchronize(s));
  PYN_CHECK(nuniq > 0, "empty pattern");
  PYN_HIP(hipMemcpy(&last, k0 + (nuniq - 1), sizeof(last), hipMemcpyDeviceToHost));
  if (last == ~0ull) --nuniq;  // the dropped (non-owned) rows
  PYN_CHECK(nuniq > 0 && nuniq < (int64_t)INT32_MAX, "pattern has %lld entries (int32 CSR limit)", (long long)nuniq);

  (void)hipFree(c->d_rowptr);
  (void)hipFree(c->d_colidx);
  c->d_rowptr = nullptr;
  c->d_colidx = nullptr;
  c->nnzb = 0;
  PYN_HIP(tcnt.alloc((c->n_owned + 1) * sizeof(int32_t)));
  int32_t* rowcnt = tcnt.as<int32_t>();
  PYN_HIP(hipMalloc((void**)&c->d_rowptr, (c->n_owned + 1) * sizeof(int32_t)));
  PYN_HIP(hipMalloc((void**)&c->d_colidx, nuniq * sizeof(int32_t)));
  PYN_HIP(hipMemsetAsync(rowcnt, 0, (c->n_owned + 1) * sizeof(int32_t), s));
  grid = (int)std::min<int64_t>((nuniq + 255) / 256, 65536);
  split_keys_kernel<<<grid, 256, 0, s>>>(k0, nuniq, c->d_colidx, rowcnt);
  tb = 0;
  PYN_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, rowcnt, c->d_rowptr, (int)(c->n_owned + 1), s));
  PYN_HIP(hipStreamSynchronize(s));
  PYN_HIP(tmp.alloc(tb));
  PYN_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, rowcnt, c->d_rowptr, (int)(c->n_owned + 1), s));
  PYN_HIP(hipEventRecord(c->ev1, s));
  PYN_HIP(hipStreamSynchronize(s));
  float ms = 0;
  PYN_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->timers[PYN_T_SYMBOLIC] = ms;
  c->nnzb = nuniq;
  return pyn_symbolic_reset_dependents(c);
}

extern "C" int pyn_csr_info(pyn_ctx* c, int64_t* n_rows, int64_t* nnz_blocks) {
  PYN_CHECK(c && c->d_rowptr, "pyn_csr_symbolic first");
  if (n_rows) *n_rows = c->n_owned;
  if (nnz_blocks) *nnz_blocks = c->nnzb;
  return PYN_OK;
}

extern "C" int pyn_csr_get(pyn_ctx* c, int32_t* rowptr, int32_t* colidx) {
  PYN_CHECK(c && c->d_rowptr, "pyn_csr_symbolic first");
  if (rowptr) PYN_HIP(hipMemcpyAsync(rowptr, c->d_rowptr, (c->n_owned + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (colidx) PYN_HIP(hipMemcpyAsync(colidx, c->d_colidx, c->nnzb * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  return PYN_OK;
}
