// SELL-64 (sliced ELLPACK, slice height = one wavefront) copy of a scalar matrix for the Krylov
// loop: lane == row, so every load of values / column indices is a contiguous 512 B / 256 B per
// wave instruction, no cross-lane reduction, and on lexicographically numbered meshes the x-gather
// of 64 consecutive rows is itself contiguous.  CSR stays the canonical storage (assembly output,
// host access); the SELL image is a solver-side acceleration structure rebuilt when values change
// (PETSc analogue: MatAssemblyEnd building the compressed-row / inode structures used by MatMult).
#include <hipcub/hipcub.hpp>

#include "pyn_internal.h"

namespace {

constexpr int PAT_MAX = 256;   // dictionary size limit (LDS: PAT_MAX * PAT_W * 4 B = 32 KB)
constexpr int PAT_W = 32;      // max row length in dictionary mode

constexpr int SH = 64;  // slice height

__global__ void sell_width_kernel(const int32_t* __restrict__ rowptr, int64_t n_rows, int64_t n_slices, int* __restrict__ w) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slices; s += (int64_t)gridDim.x * blockDim.x) {
    int64_t r0 = s * SH, r1 = r0 + SH < n_rows ? r0 + SH : n_rows;
    int m = 0;
    for (int64_t r = r0; r < r1; ++r) m = max(m, rowptr[r + 1] - rowptr[r]);
    w[s] = m;
  }
}

// one wave per slice; the slice's CSR chunk is contiguous -> staged through LDS with coalesced loads
template <bool WITH_COLS>
__global__ void __launch_bounds__(256) sell_fill_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                        const double* __restrict__ val, int64_t n_rows, int64_t n_slices,
                                                        const int64_t* __restrict__ sptr, const int* __restrict__ sw,
                                                        int maxw, double* __restrict__ sval, int32_t* __restrict__ scol) {
  extern __shared__ __align__(16) unsigned char sm[];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* lv = reinterpret_cast<double*>(sm) + (size_t)wid * SH * maxw;
  int32_t* lc = reinterpret_cast<int32_t*>(reinterpret_cast<double*>(sm) + (size_t)4 * SH * maxw) + (size_t)wid * SH * maxw;
  for (int64_t s = (int64_t)blockIdx.x * 4 + wid; s < n_slices; s += (int64_t)gridDim.x * 4) {
    const int64_t r0 = s * SH, r1 = r0 + SH < n_rows ? r0 + SH : n_rows;
    const int lo0 = rowptr[r0];
    const int cnt = rowptr[r1] - lo0;
    for (int i = lane; i < cnt; i += 64) {
      lv[i] = val[lo0 + i];
      if (WITH_COLS) lc[i] = colidx[lo0 + i];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
    const int64_t row = r0 + lane;
    int lo = 0, len = 0;
    if (row < n_rows) {
      lo = rowptr[row] - lo0;
      len = rowptr[row + 1] - rowptr[row];
    }
    const int64_t base = sptr[s];
    const int wd = sw[s];
    for (int k = 0; k < wd; ++k) {
      sval[base + (int64_t)k * SH + lane] = k < len ? lv[lo + k] : 0.0;
      if (WITH_COLS) scol[base + (int64_t)k * SH + lane] = k < len ? lc[lo + k] : (row < n_rows ? (int32_t)row : 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__device__ inline double wsum64(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <bool DOT>
__global__ void __launch_bounds__(256) sell_spmv_kernel(const int64_t* __restrict__ sptr, const int* __restrict__ sw,
                                                        const int32_t* __restrict__ scol, const double* __restrict__ sval,
                                                        const double* __restrict__ x, double* __restrict__ y, int64_t n_rows,
                                                        int64_t n_slices, const int* __restrict__ flag, double* __restrict__ part) {
  if (flag && flag[0]) return;
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double dot = 0.0;
  for (int64_t s = w0; s < n_slices; s += nw) {
    const int64_t base = sptr[s] + lane;
    const int wd = sw[s];
    const double* __restrict__ v = sval + base;
    const int32_t* __restrict__ ci = scol + base;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int k = 0;
    for (; k + 4 <= wd; k += 4) {
      const double v0 = v[(k + 0) * SH], v1 = v[(k + 1) * SH], v2 = v[(k + 2) * SH], v3 = v[(k + 3) * SH];
      const int c0 = ci[(k + 0) * SH], c1 = ci[(k + 1) * SH], c2 = ci[(k + 2) * SH], c3 = ci[(k + 3) * SH];
      a0 = fma(v0, x[c0], a0);
      a1 = fma(v1, x[c1], a1);
      a2 = fma(v2, x[c2], a2);
      a3 = fma(v3, x[c3], a3);
    }
    for (; k < wd; ++k) a0 = fma(v[k * SH], x[ci[k * SH]], a0);
    const double acc = (a0 + a1) + (a2 + a3);
    const int64_t row = s * SH + lane;
    if (row < n_rows) {
      y[row] = acc;
      if (DOT) dot += acc * x[row];
    }
  }
  if (DOT) {
    __shared__ double smd[4];
    dot = wsum64(dot);
    if (lane == 0) smd[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = smd[0] + smd[1] + smd[2] + smd[3];
  }
}


// ---- column-pattern dictionary ---------------------------------------------------------------
// On lattice-numbered meshes almost every row has one of a handful of relative column patterns
// (col - row for k = 0..len-1).  When the graph has <= PAT_MAX distinct patterns the SpMV reads a
// 4-byte pattern id per ROW instead of a 4-byte column index per ENTRY (12 -> ~8 B per nonzero);
// otherwise (unstructured numbering) the explicit column array is used.
__global__ void pat_hash_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, int64_t n_rows,
                                unsigned long long* __restrict__ h) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
    const int lo = rowptr[r], len = rowptr[r + 1] - lo;
    unsigned long long x = 1469598103934665603ull ^ (unsigned long long)len;
    for (int k = 0; k < len; ++k) {
      const long long off = (long long)colidx[lo + k] - (long long)r;
      x ^= (unsigned long long)off + 0x9e3779b97f4a7c15ull + (x << 6) + (x >> 2);
      x *= 1099511628211ull;
    }
    h[r] = x & ~(1ull << 63);  // keep clear of the all-ones pad value
  }
}

__global__ void pat_assign_kernel(const unsigned long long* __restrict__ h, int64_t n_rows,
                                  const unsigned long long* __restrict__ uniq, int nu, int32_t* __restrict__ pid,
                                  int32_t* __restrict__ rep) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long x = h[r];
    int l = 0, hgh = nu;
    while (l < hgh) {
      int m = (l + hgh) >> 1;
      if (uniq[m] < x)
        l = m + 1;
      else
        hgh = m;
    }
    pid[r] = l;
    rep[l] = (int32_t)r;  // any representative row (benign race)
  }
}

__global__ void pat_table_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                 const int32_t* __restrict__ rep, int nu, int32_t* __restrict__ tab, int32_t* __restrict__ tlen) {
  const int p = blockIdx.x;
  if (p >= nu) return;
  const int r = rep[p];
  const int lo = rowptr[r], len = rowptr[r + 1] - lo;
  if (threadIdx.x == 0) tlen[p] = len;
  for (int k = threadIdx.x; k < PAT_W; k += blockDim.x) tab[p * PAT_W + k] = k < len ? colidx[lo + k] - r : 0;
}

__global__ void pat_verify_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, int64_t n_rows,
                                  const int32_t* __restrict__ pid, const int32_t* __restrict__ tab,
                                  const int32_t* __restrict__ tlen, int* __restrict__ bad) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
    const int lo = rowptr[r], len = rowptr[r + 1] - lo;
    const int p = pid[r];
    bool ok = len == tlen[p];
    for (int k = 0; ok && k < len; ++k) ok = colidx[lo + k] == (int32_t)r + tab[p * PAT_W + k];
    if (!ok) *bad = 1;
  }
}

template <bool DOT>
__global__ void __launch_bounds__(256) sellp_spmv_kernel(const int64_t* __restrict__ sptr, const int* __restrict__ sw,
                                                         const int32_t* __restrict__ pid, const int32_t* __restrict__ tab,
                                                         int npat, const double* __restrict__ sval,
                                                         const double* __restrict__ x, double* __restrict__ y, int64_t n_rows,
                                                         int64_t n_slices, const int* __restrict__ flag,
                                                         double* __restrict__ part) {
  extern __shared__ int32_t ltab[];  // [npat][PAT_W]
  __shared__ double smd[4];
  if (flag && flag[0]) return;
  for (int i = threadIdx.x; i < npat * PAT_W; i += 256) ltab[i] = tab[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double dot = 0.0;
  for (int64_t s = w0; s < n_slices; s += nw) {
    const int64_t row = s * SH + lane;
    const int wd = sw[s];
    const double* __restrict__ v = sval + sptr[s] + lane;
    const int32_t* __restrict__ t = ltab + (row < n_rows ? pid[row] : 0) * PAT_W;
    const int64_t rb = row < n_rows ? row : 0;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int k = 0;
    for (; k + 4 <= wd; k += 4) {
      const double v0 = v[(k + 0) * SH], v1 = v[(k + 1) * SH], v2 = v[(k + 2) * SH], v3 = v[(k + 3) * SH];
      a0 = fma(v0, x[rb + t[k + 0]], a0);
      a1 = fma(v1, x[rb + t[k + 1]], a1);
      a2 = fma(v2, x[rb + t[k + 2]], a2);
      a3 = fma(v3, x[rb + t[k + 3]], a3);
    }
    for (; k < wd; ++k) a0 = fma(v[k * SH], x[rb + t[k]], a0);
    const double acc = (a0 + a1) + (a2 + a3);
    if (row < n_rows) {
      y[row] = acc;
      if (DOT) dot += acc * x[row];
    }
  }
  if (DOT) {
    dot = wsum64(dot);
    if (lane == 0) smd[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = smd[0] + smd[1] + smd[2] + smd[3];
  }
}

}  // namespace

static int build_pattern_dictionary(pyn_ctx* c, int maxw) {
  c->sell_npat = 0;
  if (maxw > PAT_W || getenv("PYNAMA_NO_PATTERNS")) return PYN_OK;
  hipStream_t s = c->stream;
  const int64_t n = c->n_owned;
  unsigned long long *h = nullptr, *hs = nullptr, *hu = nullptr;
  int64_t* d_nu = nullptr;
  void* tmp = nullptr;
  PYN_HIP(hipMalloc((void**)&h, n * sizeof(unsigned long long)));
  PYN_HIP(hipMalloc((void**)&hs, n * sizeof(unsigned long long)));
  PYN_HIP(hipMalloc((void**)&hu, n * sizeof(unsigned long long)));
  PYN_HIP(hipMalloc((void**)&d_nu, sizeof(int64_t)));
  const int grid = (int)std::min<int64_t>((n + 255) / 256, 16384);
  pat_hash_kernel<<<grid, 256, 0, s>>>(c->d_rowptr, c->d_colidx, n, h);
  size_t tb = 0;
  PYN_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, h, hs, n, 0, 64, s));
  PYN_HIP(hipMalloc(&tmp, tb));
  PYN_HIP(hipcub::DeviceRadixSort::SortKeys(tmp, tb, h, hs, n, 0, 64, s));
  PYN_HIP(hipFree(tmp));
  tmp = nullptr;
  tb = 0;
  PYN_HIP(hipcub::DeviceSelect::Unique(nullptr, tb, hs, hu, d_nu, n, s));
  PYN_HIP(hipMalloc(&tmp, tb));
  PYN_HIP(hipcub::DeviceSelect::Unique(tmp, tb, hs, hu, d_nu, n, s));
  int64_t nu = 0;
  PYN_HIP(hipMemcpyAsync(&nu, d_nu, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  PYN_HIP(hipStreamSynchronize(s));
  PYN_HIP(hipFree(tmp));
  bool ok = nu >= 1 && nu <= PAT_MAX;
  if (ok) {
    int32_t *rep = nullptr, *tlen = nullptr;
    int* bad = nullptr;
    PYN_HIP(hipMalloc((void**)&c->sell_pid, n * sizeof(int32_t)));
    PYN_HIP(hipMalloc((void**)&c->sell_tab, (size_t)PAT_MAX * PAT_W * sizeof(int32_t)));
    PYN_HIP(hipMalloc((void**)&rep, PAT_MAX * sizeof(int32_t)));
    PYN_HIP(hipMalloc((void**)&tlen, PAT_MAX * sizeof(int32_t)));
    PYN_HIP(hipMalloc((void**)&bad, sizeof(int)));
    PYN_HIP(hipMemsetAsync(bad, 0, sizeof(int), s));
    PYN_HIP(hipMemsetAsync(c->sell_tab, 0, (size_t)PAT_MAX * PAT_W * sizeof(int32_t), s));
    pat_assign_kernel<<<grid, 256, 0, s>>>(h, n, hu, (int)nu, c->sell_pid, rep);
    pat_table_kernel<<<(int)nu, 64, 0, s>>>(c->d_rowptr, c->d_colidx, rep, (int)nu, c->sell_tab, tlen);
    pat_verify_kernel<<<grid, 256, 0, s>>>(c->d_rowptr, c->d_colidx, n, c->sell_pid, c->sell_tab, tlen, bad);
    int hbad = 0;
    PYN_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, s));
    PYN_HIP(hipStreamSynchronize(s));
    PYN_HIP(hipFree(rep));
    PYN_HIP(hipFree(tlen));
    PYN_HIP(hipFree(bad));
    ok = hbad == 0;  // a hash collision would show up here: fall back to explicit columns
    if (!ok) {
      PYN_HIP(hipFree(c->sell_pid));
      PYN_HIP(hipFree(c->sell_tab));
      c->sell_pid = nullptr;
      c->sell_tab = nullptr;
    }
  }
  PYN_HIP(hipFree(h));
  PYN_HIP(hipFree(hs));
  PYN_HIP(hipFree(hu));
  PYN_HIP(hipFree(d_nu));
  c->sell_npat = ok ? (int)nu : 0;
  return PYN_OK;
}

// (re)build the SELL image of a scalar matrix; structure is shared by all matrices of the graph
int pyn_sell_ensure(pyn_ctx* c, DMat& A) {
  PYN_CHECK(A.br == 1 && A.bc == 1, "SELL image is for scalar matrices");
  hipStream_t s = c->stream;
  const int64_t n = c->n_owned;
  const int64_t ns = (n + SH - 1) / SH;
  bool need_cols = false;
  if (!c->sell_ptr) {
    int* d_w = nullptr;
    PYN_HIP(hipMalloc((void**)&d_w, ns * sizeof(int)));
    sell_width_kernel<<<(int)std::min<int64_t>((ns + 255) / 256, 4096), 256, 0, s>>>(c->d_rowptr, n, ns, d_w);
    std::vector<int> w((size_t)ns);
    PYN_HIP(hipMemcpyAsync(w.data(), d_w, ns * sizeof(int), hipMemcpyDeviceToHost, s));
    PYN_HIP(hipStreamSynchronize(s));
    std::vector<int64_t> ptr((size_t)ns + 1);
    ptr[0] = 0;
    int maxw = 0;
    for (int64_t i = 0; i < ns; ++i) {
      ptr[i + 1] = ptr[i] + (int64_t)w[i] * SH;
      maxw = std::max(maxw, w[i]);
    }
    PYN_CHECK((size_t)maxw * SH * 12 * 4 <= 160 * 1024, "rows too long (%d) for the SELL converter", maxw);
    PYN_HIP(hipMalloc((void**)&c->sell_ptr, (ns + 1) * sizeof(int64_t)));
    PYN_HIP(hipMemcpyAsync(c->sell_ptr, ptr.data(), (ns + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    PYN_HIP(hipStreamSynchronize(s));
    c->sell_w = d_w;
    c->sell_total = ptr[ns];
    c->sell_maxw = maxw;
    c->sell_ns = ns;
    PYN_HIP(hipMalloc((void**)&c->sell_col, c->sell_total * sizeof(int32_t)));
    need_cols = true;
    PYN_TRY(build_pattern_dictionary(c, maxw));
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sell_fill_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sell_fill_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  if (!A.sell_val) PYN_HIP(hipMalloc((void**)&A.sell_val, c->sell_total * sizeof(double)));
  if (!A.sell_valid || need_cols) {
    const size_t lds = (size_t)4 * SH * c->sell_maxw * 12;
    const int grid = (int)std::min<int64_t>((ns + 3) / 4, 256 * 8);
    if (need_cols)
      sell_fill_kernel<true><<<grid, 256, lds, s>>>(c->d_rowptr, c->d_colidx, A.val, n, ns, c->sell_ptr, c->sell_w, c->sell_maxw,
                                                    A.sell_val, c->sell_col);
    else
      sell_fill_kernel<false><<<grid, 256, lds, s>>>(c->d_rowptr, c->d_colidx, A.val, n, ns, c->sell_ptr, c->sell_w, c->sell_maxw,
                                                     A.sell_val, nullptr);
    PYN_HIP(hipGetLastError());
    A.sell_valid = true;
  }
  return PYN_OK;
}

int pyn_sell_spmv(pyn_ctx* c, const DMat& A, const double* x, double* y, bool dot, int* grid_out) {
  const int64_t ns = c->sell_ns;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((ns + 3) / 4, PYN_MAX_PARTIALS));
  if (c->sell_npat > 0) {
    if (dot)
      sellp_spmv_kernel<true><<<grid, 256, (size_t)c->sell_npat * PAT_W * sizeof(int32_t), c->stream>>>(c->sell_ptr, c->sell_w, c->sell_pid, c->sell_tab, c->sell_npat,
                                                           A.sell_val, x, y, c->n_owned, ns, c->d_flag, c->d_part);
    else
      sellp_spmv_kernel<false><<<grid, 256, (size_t)c->sell_npat * PAT_W * sizeof(int32_t), c->stream>>>(c->sell_ptr, c->sell_w, c->sell_pid, c->sell_tab, c->sell_npat,
                                                            A.sell_val, x, y, c->n_owned, ns, nullptr, nullptr);
  } else if (dot)
    sell_spmv_kernel<true><<<grid, 256, 0, c->stream>>>(c->sell_ptr, c->sell_w, c->sell_col, A.sell_val, x, y, c->n_owned, ns,
                                                        c->d_flag, c->d_part);
  else
    sell_spmv_kernel<false><<<grid, 256, 0, c->stream>>>(c->sell_ptr, c->sell_w, c->sell_col, A.sell_val, x, y, c->n_owned, ns,
                                                         nullptr, nullptr);
  PYN_HIP(hipGetLastError());
  if (grid_out) *grid_out = grid;
  return PYN_OK;
}

void pyn_sell_drop_structure(pyn_ctx* c) {
  (void)hipFree(c->sell_ptr);
  (void)hipFree(c->sell_w);
  (void)hipFree(c->sell_col);
  (void)hipFree(c->sell_pid);
  (void)hipFree(c->sell_tab);
  c->sell_pid = nullptr;
  c->sell_tab = nullptr;
  c->sell_npat = 0;
  c->sell_ptr = nullptr;
  c->sell_w = nullptr;
  c->sell_col = nullptr;
  c->sell_total = 0;
  c->sell_ns = 0;
}
