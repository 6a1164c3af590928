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

// Block matrices (br x bc per graph edge) are handled as their scalar expansion: scalar row
// r = i*br + p has len_i*bc entries, contiguous in the canonical value array at
// (rowptr[i]*br + p*len_i)*bc, entry k -> column colidx[rowptr[i] + k/bc]*bc + k%bc.
__global__ void sell_width_kernel(const int32_t* __restrict__ rowptr, int64_t n_rows, int br, int bc, int64_t n_slices,
                                  int* __restrict__ w) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slices; s += (int64_t)gridDim.x * blockDim.x) {
    int64_t r0 = s * SH, r1 = r0 + SH < n_rows ? r0 + SH : n_rows;
    int m = 0;
    for (int64_t r = r0; r < r1; ++r) {
      const int64_t i = r / br;
      m = max(m, (rowptr[i + 1] - rowptr[i]) * bc);
    }
    w[s] = m;
  }
}

// block matrices: direct (strided) reads, one lane per scalar row -- a one-off per assembly
template <bool WITH_COLS>
__global__ void __launch_bounds__(256) sell_fill_block_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                              const double* __restrict__ val, int64_t n_rows, int br, int bc,
                                                              int64_t n_slices, const int64_t* __restrict__ sptr,
                                                              const int* __restrict__ sw, double* __restrict__ sval,
                                                              int32_t* __restrict__ scol) {
  const int lane = threadIdx.x & 63;
  for (int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; s < n_slices; s += ((int64_t)gridDim.x * blockDim.x) >> 6) {
    const int64_t row = s * SH + lane;
    int64_t off = 0;
    int len = 0, lo = 0;
    if (row < n_rows) {
      const int64_t i = row / br;
      const int p = (int)(row - i * br);
      lo = rowptr[i];
      const int ln = rowptr[i + 1] - lo;
      off = ((int64_t)lo * br + (int64_t)p * ln) * bc;
      len = ln * bc;
    }
    const int64_t base = sptr[s];
    const int wd = sw[s];
    for (int k = 0; k < wd; ++k) {
      sval[base + (int64_t)k * SH + lane] = k < len ? val[off + k] : 0.0;
      if (WITH_COLS) scol[base + (int64_t)k * SH + lane] = k < len ? colidx[lo + k / bc] * bc + k % bc : 0;
    }
  }
}

// block matrices through LDS: the 64 scalar rows of a slice are ONE contiguous chunk of the block-CSR values (rows (i, p)
// follow each other: (rowptr[i] br + p len_i) bc), read with coalesced loads by the whole workgroup and written back as
// 512-B k-columns.  (The strided variant above touches a different cache line per lane and load: 8x the traffic.)
template <bool WITH_COLS>
__global__ void __launch_bounds__(256) sell_fill_block_lds_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                                  const double* __restrict__ val, int64_t n_rows, int br, int bc,
                                                                  int64_t n_slices, const int64_t* __restrict__ sptr,
                                                                  const int* __restrict__ sw, double* __restrict__ sval,
                                                                  int32_t* __restrict__ scol) {
  extern __shared__ __align__(16) double lv[];   // [64 * maxw]
  __shared__ int roff[SH + 1];                   // chunk-relative first entry of every scalar row of the slice
  __shared__ int rlo[SH];                        // rowptr of the row's node (columns)
  __shared__ int64_t base0;
  const int tid = threadIdx.x;
  const int64_t n_nodes = n_rows / br;
  for (int64_t s = blockIdx.x; s < n_slices; s += gridDim.x) {
    const int64_t r0 = s * SH;
    auto row_off = [&](int64_t row) -> int64_t {
      if (row >= n_rows) return (int64_t)rowptr[n_nodes] * br * bc;
      const int64_t i = row / br;
      const int p = (int)(row - i * br);
      const int lo = rowptr[i];
      return ((int64_t)lo * br + (int64_t)p * (rowptr[i + 1] - lo)) * bc;
    };
    const int64_t b0 = row_off(r0);
    if (tid <= SH) {
      roff[tid] = (int)(row_off(r0 + tid) - b0);
      if (tid < SH) rlo[tid] = r0 + tid < n_rows ? rowptr[(r0 + tid) / br] : 0;
    }
    if (tid == 0) base0 = b0;
    __syncthreads();
    const int total = roff[SH];
    const double* __restrict__ src = val + base0;
    for (int e = tid; e < total; e += 256) lv[e] = src[e];
    __syncthreads();
    const int wd = sw[s];
    const int64_t ob = sptr[s];
    for (int idx = tid; idx < wd * SH; idx += 256) {
      const int k = idx >> 6, ln = idx & 63;
      const int st = roff[ln], len = roff[ln + 1] - st;
      sval[ob + idx] = k < len ? lv[st + k] : 0.0;
      if (WITH_COLS) scol[ob + idx] = k < len ? colidx[rlo[ln] + k / bc] * bc + k % bc : 0;
    }
    __syncthreads();
  }
}

// One WAVE per slice, independent waves (single-wave workgroups: 11 fit a CU's LDS; no workgroup barrier: a slice's staging area in LDS belongs to
// its wave, and the LDS executes a wave's instructions in order).  The slice's CSR chunk is contiguous: staged through LDS
// with coalesced loads -- all of a batch in flight before the first LDS write -- and written back as 512-B k-columns.  The
// row offsets of the 64 rows come from ONE coalesced pair of loads (lane = row), so a slice costs two dependent global
// latencies (offsets, values) instead of three.
template <bool WITH_COLS>
__global__ void __launch_bounds__(64) sell_fill_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                        const double* __restrict__ val, int64_t n_rows, int64_t n_slices,
                                                        const int64_t* __restrict__ sptr, const int* __restrict__ sw,
                                                        int maxw, double* __restrict__ sval, int32_t* __restrict__ scol) {
  extern __shared__ __align__(16) unsigned char sm[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t per_wave = (size_t)SH * maxw * (WITH_COLS ? 12 : 8);
  double* lv = reinterpret_cast<double*>(sm + wv * per_wave);
  int32_t* lc = reinterpret_cast<int32_t*>(lv + (size_t)SH * maxw);
  const int nwb = blockDim.x >> 6;
  const int64_t w0 = (int64_t)blockIdx.x * nwb + wv, nw = (int64_t)gridDim.x * nwb;
  for (int64_t s = w0; s < n_slices; s += nw) {
    const int64_t row = s * SH + lane;
    const int a = rowptr[row < n_rows ? row : n_rows], b = rowptr[row + 1 < n_rows ? row + 1 : n_rows];
    const int lo0 = __builtin_amdgcn_readfirstlane(a);
    const int cnt = __builtin_amdgcn_readlane(b, 63) - lo0;
    const int lo = a - lo0, len = b - a;
    constexpr int U = 8;
    for (int i0 = 0; i0 < cnt; i0 += 64 * U) {
      double v[U];
      int32_t cc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + 64 * u + lane;
        v[u] = i < cnt ? val[lo0 + i] : 0.0;
        if (WITH_COLS) cc[u] = i < cnt ? colidx[lo0 + i] : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + 64 * u + lane;
        if (i < cnt) {
          lv[i] = v[u];
          if (WITH_COLS) lc[i] = cc[u];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the LDS writes above are ordered before the reads below
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int64_t base = sptr[s];
    const int wd = sw[s];
    for (int k = 0; k < wd; ++k) {
      sval[base + (int64_t)k * SH + lane] = k < len ? lv[lo + k] : 0.0;
      if (WITH_COLS) scol[base + (int64_t)k * SH + lane] = k < len ? lc[lo + k] : (row < n_rows ? (int32_t)row : 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // ... and these reads before the next slice's writes
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
                                                        int64_t n_slices, const int* __restrict__ flag, double* __restrict__ part, int64_t s_begin, int part_off, int64_t hole_begin, int64_t hole_len) {
  if (flag && flag[0]) return;
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double dot = 0.0;
  for (int64_t sq = s_begin + w0; sq < n_slices; sq += nw) {   // n_slices: logical end (hole removed)
    const int64_t s = sq >= hole_begin ? sq + hole_len : sq;
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
    if (threadIdx.x == 0) part[part_off + blockIdx.x] = smd[0] + smd[1] + smd[2] + smd[3];
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
                                                         double* __restrict__ part, int64_t s_begin, int part_off, int64_t hole_begin, int64_t hole_len) {
  extern __shared__ int32_t ltab[];  // [npat][PAT_W]
  __shared__ double smd[4];
  if (flag && flag[0]) return;
  // slot `npat` is an all-zero pattern for the padding lanes of the last slice (their values are
  // zero, but the gather must stay inside x: another pattern's negative offsets would not)
  for (int i = threadIdx.x; i < (npat + 1) * PAT_W; i += 256) ltab[i] = i < npat * PAT_W ? tab[i] : 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double dot = 0.0;
  for (int64_t sq = s_begin + w0; sq < n_slices; sq += nw) {   // n_slices: logical end (hole removed)
    const int64_t s = sq >= hole_begin ? sq + hole_len : sq;
    const int64_t row = s * SH + lane;
    const int wd = sw[s];
    const double* __restrict__ v = sval + sptr[s] + lane;
    const int32_t* __restrict__ t = ltab + (row < n_rows ? pid[row] : npat) * PAT_W;
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
    if (threadIdx.x == 0) part[part_off + blockIdx.x] = smd[0] + smd[1] + smd[2] + smd[3];
  }
}

// The same product straight from the CSR value array (no SELL image): one wave per 64 consecutive rows.  Their CSR entries are ONE
// contiguous run (<= 64 W doubles): it is read with fully coalesced loads (lane l takes the entries 64 j + l), parked in LDS, and
// every lane then walks its own row there (row offsets differ by the row length: odd for the 27-point stencil, i.e. free of bank
// conflicts).  The gathers of x depend only on the column pattern, not on the values: they are issued together with the value loads.
// Traffic = 8 B per stored entry (no slice padding) + 4 B pattern id + x, y per row.
// The matrix values of the lane-per-row kernels that stage a slice's run through LDS are read ONCE per product, in whole contiguous
// runs: non-temporal loads, so that they pass through L2 / the Infinity Cache without pushing out the x entries every row gathers
// (same-box A/B: 10 M-row CSR product 0.541 -> 0.522 ms, CG +3 %; 1024^2 second-order 2x2 blocks 0.548 -> 0.483-0.507 ms).  NOT in
// bcsr_spmv_kernel (several lanes per node row, three value rows per lane: 2.20 -> 2.65 ms with them) nor on the SELL images (+-0).
template <typename T_>
__device__ __forceinline__ T_ ld_stream(const T_* p) {
  return __builtin_nontemporal_load(p);
}

constexpr int CSRL_WAVES = 4;   // waves per workgroup (LDS: W x 512 B per wave): two workgroups = eight waves per CU measured best
template <int W, bool DOT>
__global__ void __launch_bounds__(64 * CSRL_WAVES) csrl_spmv_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ pid,
                                                                    const int32_t* __restrict__ tab, int npat, const double* __restrict__ val,
                                                                    const double* __restrict__ x, double* __restrict__ y, int64_t n_rows,
                                                                    int64_t n_slices, const int* __restrict__ flag, double* __restrict__ part,
                                                                    int64_t s_begin, int part_off, int64_t hole_begin, int64_t hole_len) {
  extern __shared__ double csrl_lds[];   // [CSRL_WAVES][64 W] values | [(npat + 1)][PAT_W] pattern table
  __shared__ double smd[CSRL_WAVES];
  if (flag && flag[0]) return;
  int32_t* ltab = reinterpret_cast<int32_t*>(csrl_lds + CSRL_WAVES * 64 * W);
  for (int i = threadIdx.x; i < (npat + 1) * PAT_W; i += 64 * CSRL_WAVES) ltab[i] = i < npat * PAT_W ? tab[i] : 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double* __restrict__ buf = csrl_lds + wv * 64 * W;
  const int64_t w0 = (int64_t)blockIdx.x * CSRL_WAVES + wv;
  const int64_t nw = (int64_t)gridDim.x * CSRL_WAVES;
  double dot = 0.0;
  // row offsets one slice ahead: the value loads of a slice then start without a dependent load in front of them
  int nrp0 = 0, nrp1 = 0;
  {
    const int64_t sq = s_begin + w0;
    if (sq < n_slices) {
      const int64_t r = (sq >= hole_begin ? sq + hole_len : sq) * SH + lane;
      nrp0 = rowptr[r < n_rows ? r : n_rows];
      nrp1 = rowptr[r < n_rows ? r + 1 : n_rows];
    }
  }
  for (int64_t sq = s_begin + w0; sq < n_slices; sq += nw) {   // n_slices: logical end (hole removed)
    const int64_t s = sq >= hole_begin ? sq + hole_len : sq;
    const int64_t row = s * SH + lane;
    const bool live = row < n_rows;
    const int rp0 = nrp0, rp1 = nrp1;
    if (sq + nw < n_slices) {
      const int64_t r = ((sq + nw) >= hole_begin ? sq + nw + hole_len : sq + nw) * SH + lane;
      nrp0 = rowptr[r < n_rows ? r : n_rows];
      nrp1 = rowptr[r < n_rows ? r + 1 : n_rows];
    }
    const int base = __builtin_amdgcn_readfirstlane(rp0);
    const int total = __builtin_amdgcn_readlane(rp1, 63) - base;
    const int off = rp0 - base, len = rp1 - rp0;
    const double* __restrict__ v = val + base;
    // the run of the 64 rows, coalesced; zero beyond its end
    double vr[W];
#pragma unroll
    for (int j = 0; j < W; ++j) vr[j] = (64 * j + lane < total) ? ld_stream(v + 64 * j + lane) : 0.0;
    // the row's x entries (pattern table: column = row + t[k])
    const int32_t* __restrict__ t = ltab + (live ? pid[row] : npat) * PAT_W;
    const int64_t rb = live ? row : 0;
    double xg[W];
#pragma unroll
    for (int k = 0; k < W; ++k) xg[k] = x[rb + (k < len ? t[k] : 0)];
#pragma unroll
    for (int j = 0; j < W; ++j) buf[64 * j + lane] = vr[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int k = 0; k + 1 < W; k += 2) {
      a0 = fma(k < len ? buf[off + k] : 0.0, xg[k], a0);
      a1 = fma(k + 1 < len ? buf[off + k + 1] : 0.0, xg[k + 1], a1);
    }
    if (W & 1) a0 = fma(W - 1 < len ? buf[off + W - 1] : 0.0, xg[W - 1], a0);
    const double acc = a0 + a1;
    if (live) {
      y[row] = acc;
      if (DOT) dot += acc * x[row];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // every lane has read its row before the next run overwrites the buffer
  }
  if (DOT) {
    dot = wsum64(dot);
    if (lane == 0) smd[wv] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
      double sm = 0.0;
      for (int i = 0; i < CSRL_WAVES; ++i) sm += smd[i];
      part[part_off + blockIdx.x] = sm;
    }
  }
}

// The same for BLOCK matrices whose node rows follow the dictionary (2-D second-order elements: 25 / 15 / 9 column nodes, 2x2 blocks, i.e.
// 50 / 30 / 18 entries per scalar row): lane = scalar row (i, p), the 64 rows of a slice are one contiguous run of the block-CSR values
// ((rowptr[i] BR + p len_i) BC), read coalesced, parked in LDS, walked per lane; x entries from the node pattern: (i + t[k / BC]) BC + k % BC.
// No image (the SELL image of such a matrix pads every slice to its longest row: 20 % on alternating vertex / edge rows), no refresh.
constexpr int CSRLB_WAVES = 2;
template <int W, int BR, int BC, bool GL, bool DOT>
__global__ void __launch_bounds__(64 * CSRLB_WAVES) csrlb_spmv_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ pid,
                                                                      const int32_t* __restrict__ tab, int npat, const double* __restrict__ val,
                                                                      const double* __restrict__ x, double* __restrict__ y, int64_t n_rows,
                                                                      int64_t n_slices, const int* __restrict__ flag, double* __restrict__ part,
                                                                      int64_t s_begin, int part_off, int64_t hole_begin, int64_t hole_len) {
  extern __shared__ double csrl_lds[];   // [CSRLB_WAVES][64 W] values | [(npat + 1)][PAT_W] node pattern table
  __shared__ double smd[CSRLB_WAVES];
  if (flag && flag[0]) return;
  int32_t* ltab = reinterpret_cast<int32_t*>(csrl_lds + CSRLB_WAVES * 64 * W);
  for (int i = threadIdx.x; i < (npat + 1) * PAT_W; i += 64 * CSRLB_WAVES) ltab[i] = i < npat * PAT_W ? tab[i] : 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double* __restrict__ buf = csrl_lds + wv * 64 * W;
  const int64_t w0 = (int64_t)blockIdx.x * CSRLB_WAVES + wv;
  const int64_t nw = (int64_t)gridDim.x * CSRLB_WAVES;
  const int64_t n_nodes = n_rows / BR;
  double dot = 0.0;
  for (int64_t sq = s_begin + w0; sq < n_slices; sq += nw) {   // n_slices: logical end (hole removed)
    const int64_t s = sq >= hole_begin ? sq + hole_len : sq;
    const int64_t row = s * SH + lane;
    const bool live = row < n_rows;
    const int64_t node = live ? row / BR : n_nodes;
    const int p = live ? (int)(row - node * BR) : 0;
    const int rp0 = rowptr[node], len = live ? rowptr[node + 1] - rp0 : 0;
    const int64_t off64 = ((int64_t)rp0 * BR + (int64_t)p * len) * BC;
    const int L1 = len * BC;
    // first entry of the slice and its length (lane 0 is always live; dead lanes sit at the end of the array)
    const int64_t base = ((int64_t)__builtin_amdgcn_readfirstlane((int)(off64 >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(off64 & 0xffffffff));
    const int my = (int)(off64 - base);
    const int total = __builtin_amdgcn_readlane(my + L1, 63);
    const double* __restrict__ v = val + base;
    const int32_t* __restrict__ t = ltab + (live ? pid[node] : npat) * PAT_W;
    const int64_t nb = live ? node : 0;
    double xg[W];
    if (GL) {
      // long rows (W = 50): the run goes global -> LDS without passing through registers (LDS-DMA, 16 B per lane = 1 KiB per wave
      // instruction, destination = wave base + lane x 16: exactly the contiguous image wanted); an odd last entry by one plain load
#pragma unroll
      for (int j = 0; j < (W + 1) / 2; ++j) {
        const int e = 128 * j + 2 * lane;
        if (e + 1 < total)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(v + e),
                                           (__attribute__((address_space(3))) void*)(buf + 128 * j), 16, 0, 2);   // aux 2: nt
      }
      if (BC == 2 && W % 2 == 0) {   // the two x entries of a column node by one 16-byte load
#pragma unroll
        for (int kk = 0; kk < W / 2; ++kk) {
          const double2 xv = *reinterpret_cast<const double2*>(x + (nb + (2 * kk < L1 ? t[kk] : 0)) * 2);
          xg[2 * kk] = xv.x;
          xg[2 * kk + 1] = xv.y;
        }
      } else {
#pragma unroll
        for (int k = 0; k < W; ++k) xg[k] = x[(nb + (k < L1 ? t[k / BC] : 0)) * BC + (k < L1 ? k % BC : 0)];
      }
      if ((total & 1) && lane == 0) buf[total - 1] = v[total - 1];
      __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0): the DMA writes and the one plain LDS write have landed
    } else {
      double vr[W];
#pragma unroll
      for (int j = 0; j < W; ++j) vr[j] = (64 * j + lane < total) ? ld_stream(v + 64 * j + lane) : 0.0;
#pragma unroll
      for (int k = 0; k < W; ++k) xg[k] = x[(nb + (k < L1 ? t[k / BC] : 0)) * BC + (k < L1 ? k % BC : 0)];
#pragma unroll
      for (int j = 0; j < W; ++j) buf[64 * j + lane] = vr[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int k = 0; k + 1 < W; k += 2) {
      a0 = fma(k < L1 ? buf[my + k] : 0.0, xg[k], a0);
      a1 = fma(k + 1 < L1 ? buf[my + k + 1] : 0.0, xg[k + 1], a1);
    }
    if (W & 1) a0 = fma(W - 1 < L1 ? buf[my + W - 1] : 0.0, xg[W - 1], a0);
    const double acc = a0 + a1;
    if (live) {
      y[row] = acc;
      if (DOT) dot += acc * x[row];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // every lane has read its row before the next run overwrites the buffer
  }
  if (DOT) {
    dot = wsum64(dot);
    if (lane == 0) smd[wv] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
      double sm = 0.0;
      for (int i = 0; i < CSRLB_WAVES; ++i) sm += smd[i];
      part[part_off + blockIdx.x] = sm;
    }
  }
}

// Block matrices: lane = scalar row (i,p); columns come from the NODE-level dictionary
// (col = (i + off[k / BC]) * BC + k % BC) or from the explicit expanded column array.
template <int BC, bool PAT, bool DOT>
__global__ void __launch_bounds__(256) sellb_spmv_kernel(const int64_t* __restrict__ sptr, const int* __restrict__ sw,
                                                         const int32_t* __restrict__ pid, const int32_t* __restrict__ tab,
                                                         int npat, const int32_t* __restrict__ scol,
                                                         const double* __restrict__ sval, const double* __restrict__ x,
                                                         double* __restrict__ y, int64_t n_rows, int br, int64_t n_slices,
                                                         const int* __restrict__ flag, double* __restrict__ part, int64_t s_begin, int part_off, int64_t hole_begin, int64_t hole_len) {
  extern __shared__ int32_t ltab[];  // [npat][PAT_W]
  __shared__ double smd[4];
  if (flag && flag[0]) return;
  if (PAT) {  // slot `npat` = all-zero pattern for padding lanes (see sellp_spmv_kernel)
    for (int i = threadIdx.x; i < (npat + 1) * PAT_W; i += 256) ltab[i] = i < npat * PAT_W ? tab[i] : 0;
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double dot = 0.0;
  for (int64_t sq = s_begin + w0; sq < n_slices; sq += nw) {   // n_slices: logical end (hole removed)
    const int64_t s = sq >= hole_begin ? sq + hole_len : sq;
    const int64_t row = s * SH + lane;
    const bool live = row < n_rows;
    const int64_t node = live ? row / br : 0;
    const int nk = sw[s] / BC;  // widths are multiples of BC
    const double* __restrict__ v = sval + sptr[s] + lane;
    const int32_t* __restrict__ ci = scol + (PAT ? 0 : sptr[s] + lane);
    const int32_t* __restrict__ t = ltab + (PAT ? (live ? pid[node] : npat) : 0) * PAT_W;
    double a0 = 0.0, a1 = 0.0;
    for (int kk = 0; kk < nk; ++kk) {
      const int64_t cb = PAT ? (node + t[kk]) * BC : 0;
#pragma unroll
      for (int q = 0; q < BC; ++q) {
        const int k = kk * BC + q;
        const int64_t col = PAT ? cb + q : (int64_t)ci[(int64_t)k * SH];
        if (q & 1)
          a1 = fma(v[(int64_t)k * SH], x[col], a1);
        else
          a0 = fma(v[(int64_t)k * SH], x[col], a0);
      }
    }
    const double acc = a0 + a1;
    if (live) {
      y[row] = acc;
      if (DOT) dot += acc * x[row];
    }
  }
  if (DOT) {
    dot = wsum64(dot);
    if (lane == 0) smd[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) part[part_off + blockIdx.x] = smd[0] + smd[1] + smd[2] + smd[3];
  }
}

// Block matrices (and scalar matrices with long rows) straight from the block-CSR values: G lanes per NODE row, 64 / G node rows per
// wave.  The br scalar rows of a node are one contiguous piece (rowptr[i] br + p len_i) bc of the value array: lane g of the group takes
// the entries r = g, g + G, ... of EVERY scalar row p (br coalesced loads per step), the x entry (col[r / BC] BC + r % BC) is gathered
// once per step and serves the br rows, a shuffle tree ends the row.  No image of the matrix, nothing to refresh after an assembly
// (the SELL image of a 128^3 3x3-block matrix cost 2.5 ms and 8.3 GB of traffic per assembly, and 8 % of padding in every product).
// Traffic = the block-CSR bytes: 8 B per value + 4 B per BLOCK of column index + x, y.
template <int BR, int BC, int U, bool DOT>
__global__ void __launch_bounds__(256) bcsr_spmv_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                        const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y,
                                                        int lg, int64_t n0, int64_t n1, int64_t hole_begin, int64_t hole_len,
                                                        const int* __restrict__ flag, double* __restrict__ part, int part_off,
                                                        const int32_t* __restrict__ rsel, const int32_t* __restrict__ cptr) {
  // rsel / cptr: COMPACT matrix (pyn_rhs.hip) -- logical row t is node row rsel[t], its values start at block cptr[t]
  __shared__ double smd[4];
  if (flag && flag[0]) return;
  // U entries per lane and trip: U column loads, U x gathers and U BR value loads in flight
  const int G = 1 << lg, npw = 64 >> lg;
  const int lane = threadIdx.x & 63, g = lane & (G - 1), sub = lane >> lg;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  double dot = 0.0;
  // row offsets one step ahead: a node's loads start without a dependent load in front of them
  int nlo = 0, nhi = 0, nvl = 0;
  int64_t ni = 0;
  {
    const int64_t tl = n0 + w0 * npw + sub;
    if (tl < n1) {
      const int64_t tq = tl >= hole_begin ? tl + hole_len : tl;
      ni = rsel ? rsel[tq] : tq;
      nlo = rowptr[ni];
      nhi = rowptr[ni + 1];
      nvl = rsel ? cptr[tq] : nlo;
    }
  }
  for (int64_t tb = n0 + w0 * npw; tb < n1; tb += nw * npw) {   // n1: logical end (hole removed)
    const int64_t tl = tb + sub;
    const bool live = tl < n1;
    const int64_t i = ni;
    const int lo = nlo, len = nhi - nlo, vlo = nvl;
    {
      const int64_t t2 = tl + nw * npw;
      nlo = nhi = nvl = 0;
      if (t2 < n1) {
        const int64_t tq = t2 >= hole_begin ? t2 + hole_len : t2;
        ni = rsel ? rsel[tq] : tq;
        nlo = rowptr[ni];
        nhi = rowptr[ni + 1];
        nvl = rsel ? cptr[tq] : nlo;
      }
    }
    const int L1 = len * BC;
    const double* __restrict__ v = val + (int64_t)vlo * (BR * BC);
    const int32_t* __restrict__ ci = colidx + lo;
    double acc[BR];
#pragma unroll
    for (int p = 0; p < BR; ++p) acc[p] = 0.0;
    int cn[U];   // column nodes of the next trip
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = g + u * G;
      cn[u] = r < L1 ? ci[r / BC] : 0;
    }
    for (int r0 = g; r0 < L1; r0 += U * G) {
      double a[U][BR], xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = r0 + u * G;
        const bool in = r < L1;
#pragma unroll
        for (int p = 0; p < BR; ++p) a[u][p] = in ? v[p * L1 + r] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = r0 + u * G;
        const int q = r % BC;
        xv[u] = r < L1 ? x[(int64_t)cn[u] * BC + q] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = r0 + (U + u) * G;
        cn[u] = r < L1 ? ci[r / BC] : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < BR; ++p) acc[p] = fma(a[u][p], xv[u], acc[p]);
    }
#pragma unroll
    for (int p = 0; p < BR; ++p)
      for (int o = G >> 1; o > 0; o >>= 1) acc[p] += __shfl_xor(acc[p], o, 64);
    if (live && g == 0) {
#pragma unroll
      for (int p = 0; p < BR; ++p) {
        y[i * BR + p] = acc[p];
        if (DOT) dot = fma(acc[p], x[i * BC + p], dot);
      }
    }
  }
  if (DOT) {
    dot = wsum64(dot);
    if (lane == 0) smd[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) part[part_off + blockIdx.x] = smd[0] + smd[1] + smd[2] + smd[3];
  }
}

}  // namespace

namespace {
// pattern id of a lattice row in closed form: per axis the class of the coordinate (ngl 2: first / inner / last; ngl 3: even-first /
// even-inner / even-last / odd), id = (cls_z ncls + cls_y) ncls + cls_x
__device__ __forceinline__ int lat_axis_class(int ngl, int c, int N) {
  if (ngl == 2) return c == 0 ? 0 : (c == N - 1 ? 2 : 1);
  if (c & 1) return 3;
  return c == 0 ? 0 : (c == N - 1 ? 2 : 1);
}
__global__ void pat_lattice_pid_kernel(int ngl, int dim, int NX, int NY, int NZ, int64_t n_rows, int32_t* __restrict__ pid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const int ncls = ngl == 2 ? 3 : 4;
  const int x = (int)(i % NX), y = (int)((i / NX) % NY), z = (int)(i / ((int64_t)NX * NY));
  int id = lat_axis_class(ngl, y, NY) * ncls + lat_axis_class(ngl, x, NX);
  if (dim == 3) id += lat_axis_class(ngl, z, NZ) * ncls * ncls;
  pid[i] = id;
}
// every `stride`-th row (and the last one) against its pattern
__global__ void pat_verify_sample_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, int64_t n_rows, int64_t stride,
                                         const int32_t* __restrict__ pid, const int32_t* __restrict__ tab, const int32_t* __restrict__ tlen,
                                         int* __restrict__ bad) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t r = t * stride;
  if (r >= n_rows) {
    if (r - stride >= n_rows || t == 0) return;
    r = n_rows - 1;      // the thread just past the end takes the last row
  }
  const int lo = rowptr[r], len = rowptr[r + 1] - lo;
  const int p = pid[r];
  bool ok = len == tlen[p];
  for (int k = 0; ok && k < len; ++k) ok = colidx[lo + k] == (int32_t)r + tab[p * PAT_W + k];
  if (!ok) *bad = 1;
}
}  // namespace

// The dictionary of a single-rank lattice in closed form (no hash / sort / full verification passes over the graph: 20 + 8 GB of set-up
// reads at 10 M rows): first-order hexahedra (27 patterns), first-order quadrilaterals (9), second-order quadrilaterals (16); a sample of
// rows is checked against the graph.  Slabs of a rank and every other numbering keep the hash-based construction below.
static int lattice_pattern_dictionary(pyn_ctx* c, bool* done) {
  *done = false;
  if (getenv("PYNAMA_NO_LATTICE_PATTERNS")) return PYN_OK;
  int ngl = 0, dim = 0, NX = 0, NY = 0, NZ = 1;
  if (c->lat.valid && c->lat.std_shape && c->lat.p_own0 == 0 && c->lat.n_own == c->lat.npl && c->n_ghost == 0) {
    ngl = 2;
    dim = 3;
    NX = c->lat.nx;
    NY = c->lat.ny;
    NZ = c->lat.npl;
  } else if (c->ho3.valid && c->ho3.p_own0 == 0 && c->ho3.n_own == c->ho3.npl && c->n_ghost == 0 && !(c->ho3.ngl == 3 && c->ho3.dim == 3)) {
    const Ho3Lattice& L = c->ho3;
    for (int j = 0; j < L.npl; ++j)
      if (L.P[j] != (int64_t)j * (L.dim == 3 ? L.NX * L.NY : L.NX)) return PYN_OK;
    ngl = L.ngl;
    dim = L.dim;
    NX = L.NX;
    NY = dim == 3 ? L.NY : L.npl;
    NZ = dim == 3 ? L.npl : 1;
  } else {
    return PYN_OK;
  }
  const int ncls = ngl == 2 ? 3 : 4;
  const int npat = dim == 3 ? ncls * ncls * ncls : ncls * ncls;
  // offsets of one axis per class
  auto axis_offs = [&](int cls, int* out) {
    int lo, hi;
    if (ngl == 2) {
      lo = cls == 0 ? 0 : -1;
      hi = cls == 2 ? 0 : 1;
    } else if (cls == 3) {
      lo = -1;
      hi = 1;
    } else {
      lo = cls == 0 ? 0 : -2;
      hi = cls == 2 ? 0 : 2;
    }
    int n = 0;
    for (int d = lo; d <= hi; ++d) out[n++] = d;
    return n;
  };
  std::vector<int32_t> tab((size_t)PAT_MAX * PAT_W, 0), tlen((size_t)PAT_MAX, 0);
  for (int p = 0; p < npat; ++p) {
    const int cx = p % ncls, cy = (p / ncls) % ncls, cz = p / (ncls * ncls);
    int ox[5], oy[5], oz[5] = {0, 0, 0, 0, 0};
    const int nxo = axis_offs(cx, ox), nyo = axis_offs(cy, oy), nzo = dim == 3 ? axis_offs(cz, oz) : 1;
    int k = 0;
    if (nxo * nyo * nzo > PAT_W) return PYN_OK;
    for (int iz = 0; iz < nzo; ++iz)
      for (int iy = 0; iy < nyo; ++iy)
        for (int ix = 0; ix < nxo; ++ix) tab[(size_t)p * PAT_W + k++] = (int32_t)((int64_t)oz[iz] * NX * NY + (int64_t)oy[iy] * NX + ox[ix]);
    tlen[p] = k;
  }
  hipStream_t s = c->stream;
  const int64_t n = c->n_owned;
  PYN_HIP(hipMalloc((void**)&c->sell_pid, n * sizeof(int32_t)));
  PYN_HIP(hipMalloc((void**)&c->sell_tab, (size_t)PAT_MAX * PAT_W * sizeof(int32_t)));
  DevTmp t_len, t_bad;
  PYN_HIP(t_len.alloc(PAT_MAX * sizeof(int32_t)));
  PYN_HIP(t_bad.alloc(sizeof(int)));
  PYN_HIP(hipMemsetAsync(t_bad.p, 0, sizeof(int), s));
  PYN_HIP(hipMemcpyAsync(c->sell_tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  PYN_HIP(hipMemcpyAsync(t_len.p, tlen.data(), tlen.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  pat_lattice_pid_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(ngl, dim, NX, NY, NZ, n, c->sell_pid);
  const int64_t stride = std::max<int64_t>(1, n / 65521);      // (a prime-ish count: the sample walks through every class)
  const int64_t ns = (n + stride - 1) / stride + 1;
  pat_verify_sample_kernel<<<(int)((ns + 255) / 256), 256, 0, s>>>(c->d_rowptr, c->d_colidx, n, stride, c->sell_pid, c->sell_tab,
                                                                  t_len.as<int32_t>(), t_bad.as<int>());
  int hbad = 0;
  PYN_HIP(hipMemcpyAsync(&hbad, t_bad.p, sizeof(int), hipMemcpyDeviceToHost, s));
  PYN_HIP(hipStreamSynchronize(s));
  if (hbad) {   // not the graph this closed form describes: let the general construction decide
    PYN_HIP(hipFree(c->sell_pid));
    PYN_HIP(hipFree(c->sell_tab));
    c->sell_pid = nullptr;
    c->sell_tab = nullptr;
    return PYN_OK;
  }
  c->sell_npat = npat;
  *done = true;
  return PYN_OK;
}

static int build_pattern_dictionary(pyn_ctx* c, int maxw) {
  c->sell_npat = 0;
  if (maxw > PAT_W || getenv("PYNAMA_NO_PATTERNS")) return PYN_OK;
  {
    bool done = false;
    PYN_TRY(lattice_pattern_dictionary(c, &done));
    if (done) return PYN_OK;
  }
  hipStream_t s = c->stream;
  const int64_t n = c->n_owned;
  DevTmp t_h, t_hs, t_hu, t_nu, t_tmp, t_rep, t_len, t_bad;  // scratch, released on every exit path
  PYN_HIP(t_h.alloc(n * sizeof(unsigned long long)));
  PYN_HIP(t_hs.alloc(n * sizeof(unsigned long long)));
  PYN_HIP(t_hu.alloc(n * sizeof(unsigned long long)));
  PYN_HIP(t_nu.alloc(sizeof(int64_t)));
  unsigned long long *h = t_h.as<unsigned long long>(), *hs = t_hs.as<unsigned long long>(), *hu = t_hu.as<unsigned long long>();
  int64_t* d_nu = t_nu.as<int64_t>();
  const int grid = (int)std::min<int64_t>((n + 255) / 256, 16384);
  pat_hash_kernel<<<grid, 256, 0, s>>>(c->d_rowptr, c->d_colidx, n, h);
  size_t tb = 0;
  PYN_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, h, hs, n, 0, 64, s));
  PYN_HIP(t_tmp.alloc(tb));
  PYN_HIP(hipcub::DeviceRadixSort::SortKeys(t_tmp.p, tb, h, hs, n, 0, 64, s));
  tb = 0;
  PYN_HIP(hipcub::DeviceSelect::Unique(nullptr, tb, hs, hu, d_nu, n, s));
  PYN_HIP(hipStreamSynchronize(s));
  PYN_HIP(t_tmp.alloc(tb));
  PYN_HIP(hipcub::DeviceSelect::Unique(t_tmp.p, tb, hs, hu, d_nu, n, s));
  int64_t nu = 0;
  PYN_HIP(hipMemcpyAsync(&nu, d_nu, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  PYN_HIP(hipStreamSynchronize(s));
  if (nu < 1 || nu > PAT_MAX) return PYN_OK;  // irregular numbering: explicit columns
  PYN_HIP(hipMalloc((void**)&c->sell_pid, n * sizeof(int32_t)));
  PYN_HIP(hipMalloc((void**)&c->sell_tab, (size_t)PAT_MAX * PAT_W * sizeof(int32_t)));
  PYN_HIP(t_rep.alloc(PAT_MAX * sizeof(int32_t)));
  PYN_HIP(t_len.alloc(PAT_MAX * sizeof(int32_t)));
  PYN_HIP(t_bad.alloc(sizeof(int)));
  PYN_HIP(hipMemsetAsync(t_bad.p, 0, sizeof(int), s));
  PYN_HIP(hipMemsetAsync(c->sell_tab, 0, (size_t)PAT_MAX * PAT_W * sizeof(int32_t), s));
  pat_assign_kernel<<<grid, 256, 0, s>>>(h, n, hu, (int)nu, c->sell_pid, t_rep.as<int32_t>());
  pat_table_kernel<<<(int)nu, 64, 0, s>>>(c->d_rowptr, c->d_colidx, t_rep.as<int32_t>(), (int)nu, c->sell_tab, t_len.as<int32_t>());
  pat_verify_kernel<<<grid, 256, 0, s>>>(c->d_rowptr, c->d_colidx, n, c->sell_pid, c->sell_tab, t_len.as<int32_t>(), t_bad.as<int>());
  int hbad = 0;
  PYN_HIP(hipMemcpyAsync(&hbad, t_bad.p, sizeof(int), hipMemcpyDeviceToHost, s));
  PYN_HIP(hipStreamSynchronize(s));
  if (hbad != 0) {  // a hash collision shows up here: fall back to explicit columns
    PYN_HIP(hipFree(c->sell_pid));
    PYN_HIP(hipFree(c->sell_tab));
    c->sell_pid = nullptr;
    c->sell_tab = nullptr;
    return PYN_OK;
  }
  c->sell_npat = (int)nu;
  return PYN_OK;
}

// (re)build the SELL image of a matrix; the structure (slice pointers, widths, explicit columns) is
// shared by all matrices of the same block shape, the column-pattern dictionary by all of them
// slice s needs ghost entries of x iff one of its nodes has a column >= n_owned (rows are sorted: the last one)
__global__ void slice_ghost_flag_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, int64_t n_owned,
                                        int br, int64_t ns, int* __restrict__ flags) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < ns; s += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r0 = s * SH, r1 = std::min<int64_t>(r0 + SH, n_owned * br) - 1;
    int f = 0;
    for (int64_t i = r0 / br; i <= r1 / br; ++i) f |= colidx[rowptr[i + 1] - 1] >= n_owned;
    flags[s] = f;
  }
}

// Scalar matrices whose rows follow the column-pattern dictionary (rows of at most PAT_W entries) are multiplied straight from their CSR
// values (csrl_spmv_kernel): no SELL image, no refresh after an assembly.  PYNAMA_SELL_IMAGE=1 keeps the image-based kernel.
static bool csr_product(const pyn_ctx* c, const DMat& A, const SellShape* S) {
  return A.br == 1 && A.bc == 1 && c->sell_npat > 0 && S && S->maxw <= PAT_W && !getenv("PYNAMA_SELL_IMAGE");
}

// Block matrices, and scalar matrices whose rows are too long for the dictionary (second-order elements: up to 125 entries), are
// multiplied straight from the block-CSR values as well (bcsr_spmv_kernel).  PYNAMA_BLOCK_SELL=1 keeps the SELL image (A/B, tests).
static bool bcsr_shape(int br, int bc) {
  return (br == bc && (br == 1 || br == 2 || br == 3)) || (bc == 1 && (br == 2 || br == 3)) || (br == 1 && (bc == 2 || bc == 3)) ||
         (br == 3 && bc == 2) || (br == 2 && bc == 3) || (br == 6 && bc == 3) || (br == 3 && bc == 6);
}
// 1: every product of A reads the CSR values (long rows: the SELL image would carry 20 % of padding and cost a refresh per assembly);
// 2: only one-off products do (pyn_spmv of a matrix without a current image: Krhs v, Rw w, the operators -- a refresh costs three
// products), solvers build the image, whose lane-per-row kernel is 20-25 % faster on the short rows of first-order elements; 0: never
static int bcsr_mode(const pyn_ctx* c, const DMat& A) {
  if (getenv("PYNAMA_BLOCK_SELL") || !bcsr_shape(A.br, A.bc)) return 0;
  const char* e = getenv("PYNAMA_BCSR_MIN_AVG");
  const double min_avg = e ? atof(e) : 128.0;
  const double avg = (double)c->nnzb * A.bc / (double)std::max<int64_t>(1, c->n_owned);   // entries per scalar row
  return avg >= min_avg ? 1 : 2;
}

int pyn_sell_ensure(pyn_ctx* c, DMat& A, bool solver) {
  PYN_CHECK(pyn_sell_supported(A), "no SELL kernel for block shape %dx%d", A.br, A.bc);
  hipStream_t s = c->stream;
  const int64_t n = c->n_owned * A.br;  // scalar rows
  const int64_t ns = (n + SH - 1) / SH;
  if (!c->sell_dict_built) {  // node-level dictionary, once per graph
    std::vector<int32_t> rp((size_t)c->n_owned + 1);
    PYN_HIP(hipMemcpyAsync(rp.data(), c->d_rowptr, (c->n_owned + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    PYN_HIP(hipStreamSynchronize(s));
    int maxlen = 0;
    for (int64_t i = 0; i < c->n_owned; ++i) maxlen = std::max(maxlen, rp[i + 1] - rp[i]);
    PYN_TRY(build_pattern_dictionary(c, maxlen));
    c->sell_dict_built = true;
  }
  SellShape* S = nullptr;
  for (auto& q : c->sell_shapes)
    if (q.br == A.br && q.bc == A.bc) S = &q;
  bool fresh = false;
  if (!S) {
    SellShape q;
    q.br = A.br;
    q.bc = A.bc;
    q.ns = ns;
    PYN_HIP(hipMalloc((void**)&q.w, ns * sizeof(int)));
    sell_width_kernel<<<(int)std::min<int64_t>((ns + 255) / 256, 4096), 256, 0, s>>>(c->d_rowptr, n, A.br, A.bc, ns, q.w);
    std::vector<int> w((size_t)ns);
    PYN_HIP(hipMemcpyAsync(w.data(), q.w, ns * sizeof(int), hipMemcpyDeviceToHost, s));
    PYN_HIP(hipStreamSynchronize(s));
    std::vector<int64_t> ptr((size_t)ns + 1);
    ptr[0] = 0;
    for (int64_t i = 0; i < ns; ++i) {
      ptr[i + 1] = ptr[i] + (int64_t)w[i] * SH;
      q.maxw = std::max(q.maxw, w[i]);
    }
    q.total = ptr[ns];
    PYN_HIP(hipMalloc((void**)&q.ptr, (ns + 1) * sizeof(int64_t)));
    PYN_HIP(hipMemcpyAsync(q.ptr, ptr.data(), (ns + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    PYN_HIP(hipStreamSynchronize(s));
    // explicit columns (no dictionary): allocated with the first IMAGE of this shape -- matrices multiplied from their CSR values never need them
    if (c->n_ghost > 0) {
      // interior slices (no ghost columns) can be multiplied while the halo exchange is in flight: usable when
      // they form ONE contiguous range (z-slabs: everything between the first and the last node plane)
      DevTmp tf;
      PYN_HIP(tf.alloc(ns * sizeof(int)));
      slice_ghost_flag_kernel<<<(int)std::min<int64_t>((ns + 255) / 256, 4096), 256, 0, s>>>(c->d_rowptr, c->d_colidx, c->n_owned, A.br,
                                                                                         ns, tf.as<int>());
      std::vector<int> fl((size_t)ns);
      PYN_HIP(hipMemcpyAsync(fl.data(), tf.p, ns * sizeof(int), hipMemcpyDeviceToHost, s));
      PYN_HIP(hipStreamSynchronize(s));
      int64_t i0 = 0, i1 = ns;
      while (i0 < ns && fl[i0]) ++i0;
      while (i1 > i0 && fl[i1 - 1]) --i1;
      bool contiguous = i1 > i0;
      for (int64_t i = i0; i < i1 && contiguous; ++i) contiguous = !fl[i];
      if (contiguous) {
        q.int_begin = i0;
        q.int_end = i1;
      }
    }
    c->sell_shapes.push_back(q);
    S = &c->sell_shapes.back();
    fresh = true;
  }
  A.csr_product = csr_product(c, A, S);
  A.bcsr_product = false;
  if (A.rhs_compact) {   // stored rows only, from their block-CSR values
    PYN_CHECK(bcsr_shape(A.br, A.bc), "no block-CSR product for block shape %dx%d", A.br, A.bc);
    A.csr_product = A.csrlb_product = false;
    A.bcsr_product = true;
    A.prod_ready = true;
    return PYN_OK;
  }
  A.csrlb_product = !A.csr_product && A.br == 2 && A.bc == 2 && c->sell_npat > 0 && S->maxw <= 50 && !getenv("PYNAMA_BLOCK_SELL") &&
                    !getenv("PYNAMA_NO_CSRLB");
  int bm = 0;
  if (!A.csr_product && !A.csrlb_product) {
    bm = bcsr_mode(c, A);
    if (A.br == 1 && A.bc == 1 && !(S->maxw > PAT_W && !getenv("PYNAMA_SELL_IMAGE"))) bm = 0;   // scalar rows inside the dictionary's width keep their kernels
    A.bcsr_product = bm == 1 || (bm == 2 && !solver && !(A.sell_val && A.sell_valid));
  }
  if (A.csr_product || A.csrlb_product || bm == 1) {   // never an image: the products read A.val
    if (A.sell_val) {
      (void)hipFree(A.sell_val);
      A.sell_val = nullptr;
    }
    A.sell_valid = false;
  }
  if (A.csr_product || A.csrlb_product || A.bcsr_product) {
    A.prod_ready = true;
    return PYN_OK;
  }
  if (!A.sell_val) {
    PYN_HIP(hipMalloc((void**)&A.sell_val, S->total * sizeof(double)));
    A.sell_valid = false;
  }
  const bool cols = c->sell_npat == 0 && !S->col;   // this image also writes the shape's explicit column array
  if (cols) PYN_HIP(hipMalloc((void**)&S->col, S->total * sizeof(int32_t)));
  if (!A.sell_valid || fresh || cols) {
    if (A.br == 1 && A.bc == 1 && (size_t)SH * S->maxw * 12 <= 64 * 1024) {
      const size_t lds = (size_t)SH * S->maxw * (cols ? 12 : 8);
      const int grid = (int)std::min<int64_t>(ns, 256 * 32);
      if (cols)
        sell_fill_kernel<true><<<grid, 64, lds, s>>>(c->d_rowptr, c->d_colidx, A.val, n, ns, S->ptr, S->w, S->maxw, A.sell_val, S->col);
      else
        sell_fill_kernel<false><<<grid, 64, lds, s>>>(c->d_rowptr, c->d_colidx, A.val, n, ns, S->ptr, S->w, S->maxw, A.sell_val, nullptr);
    } else if ((size_t)SH * S->maxw * sizeof(double) <= 96 * 1024 && !getenv("PYNAMA_SELL_FILL_STRIDED")) {
      const size_t lds = (size_t)SH * S->maxw * sizeof(double);
      const int grid = (int)std::min<int64_t>(ns, 256 * 8);
      PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sell_fill_block_lds_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sell_fill_block_lds_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (cols)
        sell_fill_block_lds_kernel<true><<<grid, 256, lds, s>>>(c->d_rowptr, c->d_colidx, A.val, n, A.br, A.bc, ns, S->ptr, S->w, A.sell_val, S->col);
      else
        sell_fill_block_lds_kernel<false><<<grid, 256, lds, s>>>(c->d_rowptr, c->d_colidx, A.val, n, A.br, A.bc, ns, S->ptr, S->w, A.sell_val, nullptr);
    } else {
      const int grid = (int)std::min<int64_t>((ns + 3) / 4, 256 * 16);
      if (cols)
        sell_fill_block_kernel<true><<<grid, 256, 0, s>>>(c->d_rowptr, c->d_colidx, A.val, n, A.br, A.bc, ns, S->ptr, S->w, A.sell_val, S->col);
      else
        sell_fill_block_kernel<false><<<grid, 256, 0, s>>>(c->d_rowptr, c->d_colidx, A.val, n, A.br, A.bc, ns, S->ptr, S->w, A.sell_val, nullptr);
    }
    PYN_HIP(hipGetLastError());
    A.sell_valid = true;
  }
  A.prod_ready = true;
  return PYN_OK;
}

bool pyn_sell_supported(const DMat& A) { return A.bc == 1 || A.bc == 2 || A.bc == 3 || A.bc == 6; }

template <int BC>
static int launch_block(pyn_ctx* c, const SellShape& S, const DMat& A, const double* x, double* y, bool dot, int grid,
                        int64_t s0, int64_t s1, int poff, int64_t hb, int64_t hl, hipStream_t st) {
  const int64_t n = c->n_owned * A.br;
  const size_t lds = (size_t)(c->sell_npat + 1) * PAT_W * sizeof(int32_t);
  const bool pat = c->sell_npat > 0;
  if (pat && dot)
    sellb_spmv_kernel<BC, true, true><<<grid, 256, lds, st>>>(S.ptr, S.w, c->sell_pid, c->sell_tab, c->sell_npat, nullptr,
                                                                     A.sell_val, x, y, n, A.br, s1, c->d_flag, c->d_part, s0, poff, hb, hl);
  else if (pat)
    sellb_spmv_kernel<BC, true, false><<<grid, 256, lds, st>>>(S.ptr, S.w, c->sell_pid, c->sell_tab, c->sell_npat, nullptr,
                                                                      A.sell_val, x, y, n, A.br, s1, nullptr, nullptr, s0, poff, hb, hl);
  else if (dot)
    sellb_spmv_kernel<BC, false, true><<<grid, 256, 0, st>>>(S.ptr, S.w, nullptr, nullptr, 0, S.col, A.sell_val, x, y, n,
                                                                    A.br, s1, c->d_flag, c->d_part, s0, poff, hb, hl);
  else
    sellb_spmv_kernel<BC, false, false><<<grid, 256, 0, st>>>(S.ptr, S.w, nullptr, nullptr, 0, S.col, A.sell_val, x, y, n,
                                                                     A.br, s1, nullptr, nullptr, s0, poff, hb, hl);
  return PYN_OK;
}

const SellShape* pyn_sell_shape(pyn_ctx* c, const DMat& A) {
  for (auto& q : c->sell_shapes)
    if (q.br == A.br && q.bc == A.bc) return &q;
  return nullptr;
}

// y = A x over the slices [s0, s1) on stream `st`; the fused dot partials go to d_part[poff .. poff + grid)
int pyn_sell_spmv_range(pyn_ctx* c, const DMat& A, const double* x, double* y, bool dot, int64_t s0, int64_t s1, int poff,
                        int max_grid, hipStream_t st, int* grid_out) {
  return pyn_sell_spmv_range2(c, A, x, y, dot, s0, s1, s1, s1, poff, max_grid, st, grid_out);
}

// the same over [a0, a1) U [b0, b1), a1 <= b0, in ONE launch (a rank's bottom and top boundary slices): the kernels walk the
// logical range with the hole [a1, b0) cut out
int pyn_sell_spmv_range2(pyn_ctx* c, const DMat& A, const double* x, double* y, bool dot, int64_t a0, int64_t a1, int64_t b0,
                         int64_t b1, int poff, int max_grid, hipStream_t st, int* grid_out) {
  const SellShape* S = pyn_sell_shape(c, A);
  PYN_CHECK(S && A.prod_ready, "pyn_sell_ensure first");
  PYN_CHECK(!dot || A.br == A.bc, "fused dot needs a square block shape");
  PYN_CHECK(a0 >= 0 && a0 <= a1 && a1 <= b0 && b0 <= b1 && b1 <= S->ns, "bad slice ranges");
  const int64_t hb = a1, hl = b0 - a1;        // hole in logical coordinates
  const int64_t s0 = a0, s1 = b1 - hl;        // logical range
  if (s0 == s1) {
    if (grid_out) *grid_out = 0;
    return PYN_OK;
  }
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((s1 - s0 + 3) / 4, max_grid));
  PYN_CHECK(poff + grid <= PYN_MAX_PARTIALS, "partial buffer overflow");
  if (A.bcsr_product) {   // block-CSR values, G lanes per node row; slice boundaries become node boundaries (floor: a node that straddles
                          // an interior and a boundary slice has no ghost column, either side may take it)
    auto node_of = [&](int64_t sl) { return std::min<int64_t>(c->n_owned, sl * SH / A.br); };
    int64_t na0 = node_of(a0), na1 = node_of(a1), nb0 = node_of(b0), nb1 = node_of(b1);
    if (A.rhs_compact) {   // whole products only (a right-hand-side operator is never split for a halo overlap): all stored rows
      PYN_CHECK(a0 == 0 && b1 == S->ns && a1 == b0 && !dot, "compact imposed-column matrix: whole products only");
      na0 = 0;
      na1 = nb0 = nb1 = A.c_nr;
    }
    const int64_t nhb = na1, nhl = nb0 - na1, n0 = na0, n1 = nb1 - nhl;
    if (n0 >= n1) {
      if (grid_out) *grid_out = 0;
      return PYN_OK;
    }
    const char* ge = getenv("PYNAMA_BCSR_LANES");
    int lg;
    if (ge) {
      const int gg = atoi(ge);
      lg = gg >= 64 ? 6 : gg >= 32 ? 5 : gg >= 16 ? 4 : 3;
    } else {
      const double avg = (double)c->nnzb * A.bc / (double)std::max<int64_t>(1, c->n_owned);   // entries per scalar row
      lg = avg >= 56.0 ? 4 : 3;   // measured (tools/block_spmv_case.py): 16 lanes for 81 .. 375 entries per scalar row, 8 below; 32 / 64 lanes lose
                                  // 5-10 % even on the longest rows (fewer node rows, i.e. fewer independent load streams, per wave)
    }
    const char* ue = getenv("PYNAMA_BCSR_UNROLL");
    const double avg_row = (double)c->nnzb * A.bc / (double)std::max<int64_t>(1, c->n_owned);
    const int un = ue ? atoi(ue) : (avg_row >= 24.0 ? 8 : 4);   // entries per lane and trip: whole rows in one trip where the registers allow
    const int npw = 64 >> lg;
    const char* wcu = getenv("PYNAMA_BCSR_WGS_PER_CU");
    const int64_t want = (n1 - n0 + 4 * npw - 1) / (4 * npw);
    int gridb = 1;
  // persistent waves: exactly the workgroups that are resident together (a grid of 1.6 x that capacity runs a second, half-empty round)
#define BCSR_LAUNCH_U(RR, CC, UU, DD)                                                                                                      \
  do {                                                                                                                                     \
    static int per_cu = 0;                                                                                                                 \
    if (!per_cu) {                                                                                                                         \
      PYN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bcsr_spmv_kernel<RR, CC, UU, DD>, 256, 0));                            \
      per_cu = std::max(1, std::min(per_cu, 8));                                                                                           \
    }                                                                                                                                      \
    gridb = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, max_grid), 256 * (wcu ? atoi(wcu) : per_cu)));             \
    PYN_CHECK(poff + gridb <= PYN_MAX_PARTIALS, "partial buffer overflow");                                                                \
    bcsr_spmv_kernel<RR, CC, UU, DD><<<gridb, 256, 0, st>>>(c->d_rowptr, c->d_colidx, A.val, x, y, lg, n0, n1, nhb, nhl,                   \
                                                            DD ? c->d_flag : nullptr, DD ? c->d_part : nullptr, poff,                     \
                                                            A.rhs_compact ? A.c_rsel : nullptr, A.rhs_compact ? A.c_cptr : nullptr);      \
  } while (0)
#define BCSR_LAUNCH(RR, CC, DD)                 \
  do {                                          \
    if (un == 2)                                \
      BCSR_LAUNCH_U(RR, CC, 2, DD);             \
    else if (un == 3)                           \
      BCSR_LAUNCH_U(RR, CC, 3, DD);             \
    else if (un == 8)                           \
      BCSR_LAUNCH_U(RR, CC, 8, DD);             \
    else                                        \
      BCSR_LAUNCH_U(RR, CC, 4, DD);             \
  } while (0)
#define BCSR_SQUARE(NN)          \
  do {                           \
    if (dot)                     \
      BCSR_LAUNCH(NN, NN, true); \
    else                         \
      BCSR_LAUNCH(NN, NN, false); \
  } while (0)
    const int shape = A.br * 8 + A.bc;
    switch (shape) {
      case 1 * 8 + 1: BCSR_SQUARE(1); break;
      case 2 * 8 + 2: BCSR_SQUARE(2); break;
      case 3 * 8 + 3: BCSR_SQUARE(3); break;
      case 2 * 8 + 1: BCSR_LAUNCH(2, 1, false); break;
      case 3 * 8 + 1: BCSR_LAUNCH(3, 1, false); break;
      case 1 * 8 + 2: BCSR_LAUNCH(1, 2, false); break;
      case 1 * 8 + 3: BCSR_LAUNCH(1, 3, false); break;
      case 3 * 8 + 2: BCSR_LAUNCH(3, 2, false); break;
      case 2 * 8 + 3: BCSR_LAUNCH(2, 3, false); break;
      case 6 * 8 + 3: BCSR_LAUNCH(6, 3, false); break;
      case 3 * 8 + 6: BCSR_LAUNCH(3, 6, false); break;
      default: PYN_CHECK(false, "no block-CSR product for block shape %dx%d", A.br, A.bc);
    }
#undef BCSR_LAUNCH_U
#undef BCSR_SQUARE
#undef BCSR_LAUNCH
    PYN_HIP(hipGetLastError());
    if (grid_out) *grid_out = gridb;
    return PYN_OK;
  }
  if (A.csrlb_product) {   // 2x2 blocks in dictionary mode: lane per scalar row over LDS-staged runs of the block-CSR values
    const int W = S->maxw <= 18 ? 18 : (S->maxw <= 32 ? 32 : 50);
    const size_t lds = (size_t)CSRLB_WAVES * 64 * W * sizeof(double) + (size_t)(c->sell_npat + 1) * PAT_W * sizeof(int32_t);
    const char* gcu = getenv("PYNAMA_CSRLB_WGS_PER_CU");
#define CSRLB_LAUNCH(WW, DD)                                                                                                              \
  do {                                                                                                                                    \
    static int per_cu = 0;                                                                                                                \
    if (!per_cu) {                                                                                                                        \
      PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(csrlb_spmv_kernel<WW, 2, 2, (WW > 32), DD>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)(CSRLB_WAVES * 64 * WW * sizeof(double) + (PAT_MAX + 1) * PAT_W * sizeof(int32_t))));             \
      PYN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, csrlb_spmv_kernel<WW, 2, 2, (WW > 32), DD>, 64 * CSRLB_WAVES, lds));            \
      per_cu = std::max(1, per_cu);                                                                                                       \
    }                                                                                                                                     \
    gridc = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((s1 - s0 + CSRLB_WAVES - 1) / CSRLB_WAVES, max_grid),         \
                                                        256 * (gcu ? atoi(gcu) : per_cu)));                                              \
    PYN_CHECK(poff + gridc <= PYN_MAX_PARTIALS, "partial buffer overflow");                                                               \
    csrlb_spmv_kernel<WW, 2, 2, (WW > 32), DD><<<gridc, 64 * CSRLB_WAVES, lds, st>>>(c->d_rowptr, c->sell_pid, c->sell_tab, c->sell_npat, A.val, x, y, \
                                                                          c->n_owned * A.br, s1, DD ? c->d_flag : nullptr,                \
                                                                          DD ? c->d_part : nullptr, s0, poff, hb, hl);                    \
  } while (0)
    int gridc = 1;
    if (W == 18 && dot) CSRLB_LAUNCH(18, true);
    else if (W == 18) CSRLB_LAUNCH(18, false);
    else if (W == 32 && dot) CSRLB_LAUNCH(32, true);
    else if (W == 32) CSRLB_LAUNCH(32, false);
    else if (dot) CSRLB_LAUNCH(50, true);
    else CSRLB_LAUNCH(50, false);
#undef CSRLB_LAUNCH
    PYN_HIP(hipGetLastError());
    if (grid_out) *grid_out = gridc;
    return PYN_OK;
  }
  if (A.csr_product) {   // straight from the CSR values (decided once per pyn_sell_ensure, not per launch)
    const int W = S->maxw <= 27 ? 27 : 32;
    const size_t lds = (size_t)CSRL_WAVES * 64 * W * sizeof(double) + (size_t)(c->sell_npat + 1) * PAT_W * sizeof(int32_t);
    // persistent waves: exactly the workgroups that are resident together (LDS: 160 KB per CU; 160-190 VGPRs: three / two waves per
    // SIMD), so that every wave walks the same number of slices -- a grid of 1.6 x that capacity runs 20 % longer
    const int per_cu = std::max(1, std::min((int)(163840 / (lds + 64)), 8 / CSRL_WAVES));   // eight waves per CU (ten: +3 %, twelve: +12 %)
    const char* gcu = getenv("PYNAMA_CSR_SPMV_WGS_PER_CU");
    const int resident = 256 * (gcu ? atoi(gcu) : per_cu);
    const int gridc = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((s1 - s0 + CSRL_WAVES - 1) / CSRL_WAVES, max_grid), resident));
#define CSRL_LAUNCH(WW, DD)                                                                                                             \
  do {                                                                                                                                  \
    static bool attr = false;                                                                                                           \
    if (!attr) {                                                                                                                        \
      PYN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(csrl_spmv_kernel<WW, DD>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)(CSRL_WAVES * 64 * WW * sizeof(double) + (PAT_MAX + 1) * PAT_W * sizeof(int32_t))));             \
      attr = true;                                                                                                                      \
    }                                                                                                                                   \
    csrl_spmv_kernel<WW, DD><<<gridc, 64 * CSRL_WAVES, lds, st>>>(c->d_rowptr, c->sell_pid, c->sell_tab, c->sell_npat, A.val, x, y,     \
                                                                  c->n_owned, s1, DD ? c->d_flag : nullptr, DD ? c->d_part : nullptr,  \
                                                                  s0, poff, hb, hl);                                                    \
  } while (0)
    PYN_CHECK(poff + gridc <= PYN_MAX_PARTIALS, "partial buffer overflow");
    if (W == 27 && dot) CSRL_LAUNCH(27, true);
    else if (W == 27) CSRL_LAUNCH(27, false);
    else if (dot) CSRL_LAUNCH(32, true);
    else CSRL_LAUNCH(32, false);
#undef CSRL_LAUNCH
    PYN_HIP(hipGetLastError());
    if (grid_out) *grid_out = gridc;
    return PYN_OK;
  }
  if (A.br == 1 && A.bc == 1) {  // scalar fast paths
    if (c->sell_npat > 0) {
      const size_t lds = (size_t)(c->sell_npat + 1) * PAT_W * sizeof(int32_t);
      if (dot)
        sellp_spmv_kernel<true><<<grid, 256, lds, st>>>(S->ptr, S->w, c->sell_pid, c->sell_tab, c->sell_npat, A.sell_val, x, y,
                                                        c->n_owned, s1, c->d_flag, c->d_part, s0, poff, hb, hl);
      else
        sellp_spmv_kernel<false><<<grid, 256, lds, st>>>(S->ptr, S->w, c->sell_pid, c->sell_tab, c->sell_npat, A.sell_val, x, y,
                                                         c->n_owned, s1, nullptr, nullptr, s0, poff, hb, hl);
    } else if (dot) {
      sell_spmv_kernel<true><<<grid, 256, 0, st>>>(S->ptr, S->w, S->col, A.sell_val, x, y, c->n_owned, s1, c->d_flag, c->d_part, s0, poff, hb, hl);
    } else {
      sell_spmv_kernel<false><<<grid, 256, 0, st>>>(S->ptr, S->w, S->col, A.sell_val, x, y, c->n_owned, s1, nullptr, nullptr, s0, poff, hb, hl);
    }
  } else if (A.bc == 1) {
    PYN_TRY(launch_block<1>(c, *S, A, x, y, dot, grid, s0, s1, poff, hb, hl, st));
  } else if (A.bc == 2) {
    PYN_TRY(launch_block<2>(c, *S, A, x, y, dot, grid, s0, s1, poff, hb, hl, st));
  } else if (A.bc == 3) {
    PYN_TRY(launch_block<3>(c, *S, A, x, y, dot, grid, s0, s1, poff, hb, hl, st));
  } else {
    PYN_TRY(launch_block<6>(c, *S, A, x, y, dot, grid, s0, s1, poff, hb, hl, st));
  }
  PYN_HIP(hipGetLastError());
  if (grid_out) *grid_out = grid;
  return PYN_OK;
}

int pyn_sell_spmv(pyn_ctx* c, const DMat& A, const double* x, double* y, bool dot, int* grid_out) {
  const SellShape* S = pyn_sell_shape(c, A);
  PYN_CHECK(S, "pyn_sell_ensure first");
  return pyn_sell_spmv_range(c, A, x, y, dot, 0, S->ns, 0, PYN_MAX_PARTIALS, c->stream, grid_out);
}

void pyn_sell_drop_structure(pyn_ctx* c) {
  for (auto& q : c->sell_shapes) {
    (void)hipFree(q.ptr);
    (void)hipFree(q.w);
    (void)hipFree(q.col);
  }
  c->sell_shapes.clear();
  (void)hipFree(c->sell_pid);
  (void)hipFree(c->sell_tab);
  c->sell_pid = nullptr;
  c->sell_tab = nullptr;
  c->sell_npat = 0;
  c->sell_dict_built = false;
}
